"""Seeded random chains through the C++ host mirror on the CPU stand-in (tests/cpp/fake_device.cpp: test infrastructure) against the oracle:
sources of one or several spans with formats of their own (whole frames or a cut last frame), span kinds, random adapter chains and random
block sizes.  What the fixed cases of tests/test_host_mirror.py pin one by one, in combinations nobody wrote down: span readers, the planner
of `uniform` (sample carries, cut tails), format marks through the block pump, adapters that re-make their state at a span boundary.
No GPU: the fake device runs the C ABI's operations in the oracle's order, so the host side is all that is under test here."""
import os
import subprocess

import numpy as np
import pytest

import test_host_mirror as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "cpp", "host_mirror_test_fake")
TOL = 1e-5
RATES = [8000, 22050, 44100, 48000, 96000]


def _ops_any_format(rng, n_ops, channel_counts):
    """Adapters whose chains follow an upstream that changes its format (include/rodio_hip.hpp: Stage::fmt != 2).  A filter in front of a
    change of the channel COUNT is refused (blt.rs:128), so filters only go behind a `uniform` when the parts differ in channels."""
    ops, fixed = [], len(set(channel_counts)) == 1
    for _ in range(n_ops):
        kind = rng.choice(["amplify", "filter", "limit", "agc", "uniform", "uniform"])
        if kind == "amplify":
            ops.append(f"amplify:{rng.choice([0.5, 0.7, 1.25])}")
        elif kind == "filter":
            if fixed:
                ops.append(f"{rng.choice(['low_pass', 'high_pass'])}:{rng.choice([800, 1000, 3000])}")
        elif kind == "limit":
            ops.append("limit")
        elif kind == "agc":
            if "agc" not in ops:
                ops.append("agc")
        else:
            ops.append(f"uniform:{rng.choice([1, 2, 3, 6])}:{rng.choice(RATES)}")
            fixed = True
    return ops or ["amplify:0.5"]


def _oracle_chain(O, src, ops):
    for op in ops:
        t = op.split(":")
        if t[0] == "amplify":
            src = src.amplify(float(np.float32(float(t[1]))))
        elif t[0] == "low_pass":
            src = src.low_pass(int(t[1]))
        elif t[0] == "high_pass":
            src = src.high_pass(int(t[1]))
        elif t[0] == "limit":
            src = src.limit()
        elif t[0] == "agc":
            src = src.automatic_gain_control()
        elif t[0] == "uniform":
            src = O.UniformSourceIterator(src, int(t[1]), int(t[2]))
        else:
            raise AssertionError(op)
    return src


def _tolerance(ops, ref):
    inexact = [op for op in ops if op.startswith(("low_pass", "high_pass", "limit", "agc"))]
    if not inexact:
        return None
    tol = 2 * TOL * max(1.0, float(np.max(np.abs(ref))) if len(ref) else 1.0)
    if "agc" in ops and len(inexact) > 1:
        tol *= 8  # (the AGC's gain -- up to 7 -- multiplies what a filter or the limiter in front of it left)
    return tol


def _sequence_case(O, tmp_path, seed, exe):
    rng = np.random.default_rng(77000 + seed)
    n_parts = int(rng.integers(1, 4))
    parts = []
    for k in range(n_parts):
        ch = int(rng.choice([1, 2, 2, 2, 3, 6]))
        rate = int(rng.choice(RATES))
        frames = int(rng.integers(1, 9000))
        cut = int(rng.integers(0, ch)) if rng.random() < 0.3 else 0  # a part that ends inside a frame
        parts.append((M.rnd(77000 + 100 * seed + k, frames * ch + cut, 0.5), ch, rate))
    ops = _ops_any_format(rng, int(rng.integers(1, 4)), [p[1] for p in parts])
    block = int(rng.choice([64, 777, 4096, 16384]))
    M._write_seq(tmp_path, 0, parts)
    r = subprocess.run([exe, "chain", str(tmp_path), str(parts[0][1]), str(parts[0][2]), str(block)] + ops, capture_output=True, text=True, timeout=300)
    what = (seed, [(len(x), c, rt) for x, c, rt in parts], ops, block)
    if seed in REFUSED:  # loud refusals are part of the contract -- where they are expected
        assert r.returncode == 1 and "unsupported" in r.stderr.lower() and REFUSED[seed] in r.stderr, (what, r.stderr)
        return
    assert r.returncode == 0, (what, r.stderr)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    ref = _oracle_chain(O, O.SeqSource(parts), ops).collect()
    assert len(got) == len(ref), (what, len(got), len(ref))
    tol = _tolerance(ops, ref)
    if tol is None:
        assert np.array_equal(got, ref), (what, int(np.argmax(got != ref)))
    else:
        assert float(np.max(np.abs(got - ref))) <= tol if len(ref) else True, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


def _one_source_case(O, tmp_path, seed, exe):
    rng = np.random.default_rng(88000 + seed)
    ch = int(rng.choice([1, 2, 2, 3, 5, 6]))
    rate = int(rng.choice(RATES))
    n = int(rng.integers(1, 30000))
    kind = str(rng.choice(["test", "buffer", f"spans:{int(rng.choice([37, 1000, 2304, 32768]))}"]))
    if kind in ("test", "buffer") or rng.random() < 0.5:
        # a Source MUST end on a frame (source/mod.rs:169-178; the mirror drops what a continuous source emits beyond its last whole frame);
        # spans that cut frames are what `.min(32768)` and queues of sounds make, so half of the spanned cases keep their odd lengths
        n -= n % ch
        n = max(n, ch)
    x = M.rnd(88000 + seed, n, 0.5)
    ops = _ops_any_format(rng, int(rng.integers(1, 4)), [ch])
    block = int(rng.choice([64, 777, 4096, 16384]))
    x.tofile(tmp_path / "src_0.f32")
    r = subprocess.run([exe, "chain", str(tmp_path), str(ch), str(rate), str(block)] + ops, capture_output=True, text=True, timeout=300, env=dict(os.environ, RH_TEST_SOURCE=kind))
    what = (seed, (n, ch, rate, kind), ops, block)
    assert r.returncode == 0, (what, r.stderr)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    ref = _oracle_chain(O, M._span_source(O, kind, x, ch, rate), ops).collect()
    assert len(got) == len(ref), (what, len(got), len(ref))
    tol = _tolerance(ops, ref)
    if tol is None:
        assert np.array_equal(got, ref), (what, int(np.argmax(got != ref)))
    else:
        assert (float(np.max(np.abs(got - ref))) <= tol) if len(ref) else True, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


# seeds 0..: the suite's; the ones behind them: cases that failed while this file was written (a cut frame in a middle span in front of the
# limiter / a filter; the source's span ending inside a frame of the iterator's chain, with an AGC in front of it; two `uniform`s in one chain;
# a new sample rate inside a frame in front of a filter -- refused)
SEQ_SEEDS = list(range(48)) + [162, 248, 477]
# (162, 248, 477: a span ends inside a frame and the next one brings another sample rate, in front of a filter -- a frame with two sets of
# coefficients: refused for a while, mirrored since -- GpuSource::commit_open_frame)
REFUSED = {}
ONE_SEEDS = list(range(32))


@pytest.mark.parametrize("seed", SEQ_SEEDS)
def test_random_chain_over_a_sequence_of_formats(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _sequence_case(O, tmp_path, seed, FAKE)


@pytest.mark.parametrize("seed", ONE_SEEDS)
def test_random_chain_over_one_source_of_any_span_kind(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _one_source_case(O, tmp_path, seed, FAKE)


# ... and the same cases through the real library (the kernels under the host logic): a third of them, the GPU suite has its budget
@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEQ_SEEDS[::3] + [162, 248, 477])
def test_gpu_random_chain_over_a_sequence_of_formats(O, tmp_path, seed):
    _sequence_case(O, tmp_path, seed, M.EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", ONE_SEEDS[::3])
def test_gpu_random_chain_over_one_source_of_any_span_kind(O, tmp_path, seed):
    _one_source_case(O, tmp_path, seed, M.EXE)


# ------------------------------------------------------------------ ... and into a GpuMixer ----
def _mixer_case(O, tmp_path, seed, exe):
    """`mixer.add(chain(source))` for a handful of random sources -- plain ones of any span kind, queues of sounds of different formats, with and
    without adapter chains, chains handed over on the device or through the host -- into mixers of 1, 2 and 6 channels: the samples of rodio's
    Mixer over UniformSourceIterator(chain(source).amplify(gain))."""
    rng = np.random.default_rng(99000 + seed)
    S = int(rng.integers(1, 6))
    mixer_ch = int(rng.choice([1, 2, 2, 6]))
    to_rate = int(rng.choice([22050, 44100, 48000]))
    block = int(rng.choice([777, 4096, 20000]))
    on_device = bool(rng.integers(0, 2))
    kind = str(rng.choice(["test", "buffer", "mixed", "spans:2304", "spans:1000"]))
    lines, adds = [], []
    for i in range(S):
        gain = float(np.float32(rng.choice([0.5, 0.8, 1.0, 1.2])))
        seq = rng.random() < 0.4
        if seq:
            parts = []
            for k in range(int(rng.integers(1, 4))):
                ch = int(rng.choice([1, 2, 2, 6]))
                parts.append((M.rnd(99000 + 1000 * seed + 10 * i + k, int(rng.integers(1, 6000)) * ch, 0.2), ch, int(rng.choice(RATES))))
            M._write_seq(tmp_path, i, parts)
            ch0, rate0, chans = parts[0][1], parts[0][2], [p[1] for p in parts]
            make = lambda parts=parts: O.SeqSource(parts)
        else:
            ch0, rate0 = int(rng.choice([1, 2, 2, 6])), int(rng.choice(RATES))
            x = M.rnd(99000 + 1000 * seed + 10 * i, int(rng.integers(1, 15000)) * ch0, 0.2)
            x.tofile(tmp_path / f"src_{i}.f32")
            chans = [ch0]
            make = lambda x=x, ch0=ch0, rate0=rate0, i=i: M._span_source(O, kind, x, ch0, rate0, i)
        ops = _ops_any_format(rng, int(rng.integers(1, 3)), chans) if rng.random() < 0.5 else []
        # `mixer.add(source.low_pass(f))` with the filter given to the mixer (GpuMixer::add(src, gain, filter): the fused path's filter classes);
        # mixers of one and two channels only (a wide mixer takes chains)
        fk, ff = (int(rng.integers(0, 2)), int(rng.choice([300, 1000, 3000]))) if (mixer_ch <= 2 and rng.random() < 0.4) else (-1, 0)
        lines.append(f"{ch0} {rate0} {gain} {fk} {ff} {','.join(ops) if ops else '-'}\n")
        adds.append((make, ops, gain, fk, ff))
    (tmp_path / "spec.txt").write_text("".join(lines))
    r = subprocess.run([exe, "chainmix", str(tmp_path), str(S), str(mixer_ch), str(to_rate), str(block), "1" if on_device else "0"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RH_TEST_SOURCE=kind))
    what = (seed, S, mixer_ch, to_rate, block, on_device, kind, lines)
    assert r.returncode == 0, (what, r.stderr)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    m = O.Mixer(mixer_ch, to_rate)
    for make, ops, gain, fk, ff in adds:
        u = O.UniformSourceIterator(_oracle_chain(O, make(), ops).amplify(gain), mixer_ch, to_rate)
        m.add(u.low_pass(ff) if fk == 0 else u.high_pass(ff) if fk == 1 else u)
    ref = m.collect()
    assert len(got) == len(ref), (what, len(got), len(ref))
    if len(ref):
        tol = 2 * TOL * max(1.0, float(np.max(np.abs(ref)))) * (8 if any("agc" in l for l in lines) else 1)
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


MIX_SEEDS = list(range(40))


@pytest.mark.parametrize("seed", MIX_SEEDS)
def test_random_sources_into_a_mixer(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _mixer_case(O, tmp_path, seed, FAKE)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", MIX_SEEDS[::3])
def test_gpu_random_sources_into_a_mixer(O, tmp_path, seed):
    _mixer_case(O, tmp_path, seed, M.EXE)


# ------------------------------------------------------------------ ... and every adapter of the mirror over a continuous source ----
def _full_ops(rng, ch, n_ops, dither=True):
    """The whole vocabulary of the test driver (tests/cpp/host_mirror_test.cpp: apply_op) in random order; the channel count is tracked
    because channel_volume / channels / spatial / uniform change it."""
    ops = []
    for _ in range(n_ops):
        kind = str(rng.choice(["amplify", "filter", "limit", "agc", "uniform", "reverb", "take", "delay", "fade_in", "fade_out", "distortion", "channel_volume", "channels", "spatial"] + (["dither"] if dither else [])))
        if kind == "amplify":
            ops.append(f"amplify:{rng.choice([0.5, 0.7, 1.25])}")
        elif kind == "filter":
            ops.append(f"{rng.choice(['low_pass', 'high_pass'])}:{rng.choice([800, 1000, 3000])}")
        elif kind in ("limit", "agc"):
            if kind not in ops:
                ops.append(kind)
        elif kind == "uniform":
            ch = int(rng.choice([1, 2, 3, 6]))
            ops.append(f"uniform:{ch}:{rng.choice(RATES)}")
        elif kind == "reverb":
            ops.append(f"reverb:{int(rng.choice([1000000, 20833333, 150000000]))}:{rng.choice([0.3, 0.5])}")
        elif kind == "take":
            ops.append(f"take:{int(rng.choice([1000000, 50000000, 300000007]))}:{int(rng.integers(0, 2))}")
        elif kind == "delay":
            ops.append(f"delay:{int(rng.choice([1000000, 10000000, 300000000]))}")
        elif kind in ("fade_in", "fade_out"):
            ops.append(f"{kind}:{int(rng.choice([1000000, 100000000]))}")
        elif kind == "distortion":
            ops.append(f"distortion:{rng.choice([2.0, 4.0])}:{rng.choice([0.3, 0.6])}")
        elif kind == "channel_volume":
            to = int(rng.choice([1, 2, 4]))
            ops.append("channel_volume:" + ",".join(str(v) for v in rng.choice([0.25, 0.5, 1.0, 0.75], to)))
            ch = to
        elif kind == "channels":
            ch = int(rng.choice([1, 2, 3, 6]))
            ops.append(f"channels:{ch}")
        elif kind == "dither":
            ops.append(f"dither:{int(rng.choice([16, 24]))}:{int(rng.integers(0, 4))}:{int(rng.integers(0, 100))}")
        elif kind == "spatial" and ch == 1:  # (rodio's Spatial takes a mono source, spatial.rs:19-24 ... the driver's op is channel_volume with the two ear gains)
            ops.append("spatial")
            ch = 2
    return ops or ["amplify:0.5"]


def _oracle_full(O, src, ops):
    algos = ["GPDF", "HighPass", "RPDF", "TPDF"]
    for op in ops:
        t = op.split(":")
        if t[0] in ("amplify", "low_pass", "high_pass", "limit", "agc", "uniform"):
            src = _oracle_chain(O, src, [op])
        elif t[0] == "reverb":
            src = src.reverb(int(t[1]), float(t[2]))
        elif t[0] == "take":
            src = src.take_duration(int(t[1]), t[2] == "1")
        elif t[0] == "delay":
            src = src.delay(int(t[1]))
        elif t[0] == "fade_in":
            src = src.fade_in(int(t[1]))
        elif t[0] == "fade_out":
            src = src.fade_out(int(t[1]))
        elif t[0] == "distortion":
            src = src.distortion(float(t[1]), float(t[2]))
        elif t[0] == "channel_volume":
            src = O.ChannelVolume(src, [float(v) for v in t[1].split(",")])
        elif t[0] == "channels":
            src = O.ChannelCountConverter(src, src.channels(), int(t[1]))
        elif t[0] == "dither":
            src = src.dither(int(t[1]), algos[int(t[2])], int(t[3]))
        elif t[0] == "spatial":
            src = O.Spatial(src, [0.5, 0.0, 1.0], [-1.0, 0.0, 0.0], [1.0, 0.0, 0.0])
        else:
            raise AssertionError(op)
    return src


def _full_case(O, tmp_path, seed, exe, dither=True, partial=False):
    rng = np.random.default_rng(66000 + seed)
    ch = int(rng.choice([1, 1, 2, 2, 3, 6]) if not partial else rng.choice([2, 2, 3, 6]))
    rate = int(rng.choice(RATES))
    n = int(rng.integers(1, 12000)) * ch + (int(rng.integers(1, ch)) if partial else 0)
    x = M.rnd(66000 + seed, n, 0.5)
    ops = _full_ops(rng, ch, int(rng.integers(1, 5)), dither)
    block = int(rng.choice([64, 777, 4096, 16384]))
    x.tofile(tmp_path / "src_0.f32")
    r = subprocess.run([exe, "chain", str(tmp_path), str(ch), str(rate), str(block)] + ops, capture_output=True, text=True, timeout=300)
    what = (seed, (n, ch, rate), ops, block)
    if r.returncode == 1 and "channel_volume" in r.stderr and "unsupported" in r.stderr.lower():
        # (a channel_volume whose input ends inside a frame -- a cut last frame, a reverb or a delay by no whole number of frames in front of
        # it -- in front of a converter: rodio's ChannelVolume, polled again after its None, returns a frame of its stale sum)
        pytest.skip(f"refused: {r.stderr.strip()[:160]}")
    assert r.returncode == 0, (what, r.stderr)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    ref_src = _oracle_full(O, O.TestSource(x, ch, rate), ops)
    ref = ref_src.collect()
    fmt = (tmp_path / "format.txt").read_text().split()
    assert (int(fmt[0]), int(fmt[1])) == (ref_src.channels(), ref_src.sample_rate()), what
    assert len(got) == len(ref), (what, len(got), len(ref))
    tol = _tolerance(ops, ref)
    gpdf = [i for i, op in enumerate(ops) if op.startswith("dither:") and op.split(":")[2] == "0"]
    if tol is None and gpdf:  # Gaussian dither: logf / cosf differ in the last bit between the device and the host (tests/test_gpu_parity.py: 1e-7) ...
        amp = 1.0
        for op in ops[gpdf[0] + 1:]:  # ... times what stands behind it
            amp *= float(op.split(":")[1]) if op.startswith(("amplify", "distortion")) else 1.0
        tol = 1e-7 * max(1.0, amp) * len(gpdf)
    if tol is None:
        assert np.array_equal(got, ref), (what, int(np.argmax(got != ref)))
    elif len(ref):
        if any(op.startswith("distortion") for op in ops):
            tol *= 8  # (its gain of 2 or 4 in front of the clip)
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


# (dither's noise is the library's contract, not rodio's -- rodio seeds from entropy --: the oracle and the stand-in device both state it, rodio_hip.h)
FULL_SEEDS = list(range(int(os.environ.get("RH_FUZZ_FULL", "40"))))


@pytest.mark.parametrize("seed", list(range(60)))
def test_random_chain_of_any_adapters(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _full_case(O, tmp_path, seed, FAKE)


@pytest.mark.parametrize("seed", list(range(40)))
def test_random_chain_over_a_continuous_source_that_ends_inside_a_frame(O, tmp_path, seed):
    """A source that reports no spans and ends inside a frame all the same (source/mod.rs:169-178 asks for whole frames; a truncated file
    decodes to one): the adapters carry the open frame as they do for a span that ends inside one.  Was: refused."""
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _full_case(O, tmp_path, seed, FAKE, dither=False, partial=True)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", FULL_SEEDS)
def test_gpu_random_chain_of_any_adapters(O, tmp_path, seed):
    _full_case(O, tmp_path, seed, M.EXE)


# ------------------------------------------------------------------ ... and sources that join a mixer that is already running ----
def _late_case(O, tmp_path, seed, exe):
    """`Mixer::add` on a mixer whose consumer has pulled `pull_first` samples (mixer.rs:175-183: the new source starts at the next frame
    boundary of the mix): sources of any layout and span kind, mixers of 1 / 2 / 6 channels, joins in the first block, deep into the
    stream, and behind its end."""
    rng = np.random.default_rng(55000 + seed)
    S0, S1 = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    mixer_ch = int(rng.choice([1, 2, 2, 6]))
    to_rate = int(rng.choice([22050, 44100, 48000]))
    block = int(rng.choice([777, 4096, 7000]))
    kind = str(rng.choice(["test", "buffer", "mixed", "spans:2304"]))
    spec, xs = [], []
    for i in range(S0 + S1):
        ch, rate = int(rng.choice([1, 2, 2, 6])), int(rng.choice(RATES))
        gain = float(np.float32(rng.choice([0.5, 0.8, 1.0])))
        x = M.rnd(55000 + 100 * seed + i, int(rng.integers(1, 12000)) * ch, 0.2)
        x.tofile(tmp_path / f"src_{i}.f32")
        spec.append((ch, rate, gain))
        xs.append(x)
    (tmp_path / "spec.txt").write_text("".join(f"{c} {r} {g}\n" for c, r, g in spec))
    chain = lambda i: O.UniformSourceIterator(M._span_source(O, kind, xs[i], spec[i][0], spec[i][1], i).amplify(spec[i][2]), mixer_ch, to_rate)
    m = O.Mixer(mixer_ch, to_rate)
    for i in range(S0):
        m.add(chain(i))
    # how long the first sources play decides where a join can land: inside the stream, or behind its end
    total0 = len(O_collect_copy(O, kind, xs, spec, S0, mixer_ch, to_rate))
    pull_first = int(rng.integers(0, max(1, int(total0 * 1.2)) + 1))
    ref = []
    for _ in range(pull_first):
        v = m.next()
        if v is None:
            break
        ref.append(v)
    for i in range(S0, S0 + S1):
        m.add(chain(i))
    # an ended mixer that was given a new source answers None until its channel position is back at 0 (mixer.rs:120-136): the consumer keeps asking
    nones, v = 0, None
    while nones < 16:
        v = m.next()
        if v is not None:
            break
        nones += 1
    if v is not None:
        ref = np.concatenate([np.asarray(ref + [v], dtype=np.float32), m.collect()])
    else:
        ref = np.asarray(ref, dtype=np.float32)
    r = subprocess.run([exe, "latewide", str(tmp_path), str(S0), str(S1), str(mixer_ch), str(to_rate), str(block), str(pull_first)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RH_TEST_SOURCE=kind))
    what = (seed, S0, S1, mixer_ch, to_rate, block, kind, pull_first, total0, spec, [len(x) for x in xs])
    assert r.returncode == 0, (what, r.stderr)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert len(got) == len(ref), (what, len(got), len(ref))
    # (sources of one input rate share a fused handle, so three sources of two rates are summed as s0 + (s1 + s2) where rodio's Mixer folds
    # ((s0 + s1) + s2), mixer.rs:185-198: an ulp, inside north_star's 1e-5 for the mix)
    assert float(np.max(np.abs(got - ref))) <= TOL if len(ref) else True, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))
    assert int((tmp_path / "nones.txt").read_text()) == nones, what


def O_collect_copy(O, kind, xs, spec, S0, mixer_ch, to_rate):
    m = O.Mixer(mixer_ch, to_rate)
    for i in range(S0):
        m.add(O.UniformSourceIterator(M._span_source(O, kind, xs[i], spec[i][0], spec[i][1], i).amplify(spec[i][2]), mixer_ch, to_rate))
    return m.collect()


LATE_SEEDS = list(range(40))


@pytest.mark.parametrize("seed", LATE_SEEDS)
def test_random_late_joins(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _late_case(O, tmp_path, seed, FAKE)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", LATE_SEEDS[::3])
def test_gpu_random_late_joins(O, tmp_path, seed):
    _late_case(O, tmp_path, seed, M.EXE)


# ------------------------------------------------------------------ ... and every adapter over sources whose spans cut frames ----
def _full_span_case(O, tmp_path, seed, exe):
    """The whole vocabulary (without dither) over SamplesBuffers and decoder-like sources whose spans of 37 / 1000 / 2304 samples end inside
    frames of 2, 3 and 6 channels, some of them ending inside a frame: what the adapters that regroup the flat stream (ChannelCountConverter,
    ChannelVolume) and the ones that count positions (take_duration, the ramps) make of blocks that stop inside a frame.  A combination the
    mirror does not take is refused loudly (skipped here); a silent difference is a failure."""
    rng = np.random.default_rng(44000 + seed)
    ch = int(rng.choice([1, 2, 2, 3, 6]))
    rate = int(rng.choice(RATES))
    n = int(rng.integers(1, 9000)) * ch
    kind = str(rng.choice(["buffer", "spans:1000", "spans:2304", "spans:37"]))
    if rng.random() < 0.4:
        n += int(rng.integers(0, ch))  # a source that ends inside a frame
    x = M.rnd(44000 + seed, n, 0.5)
    ops = _full_ops(rng, ch, int(rng.integers(1, 4)), False)
    block = int(rng.choice([64, 777, 4096]))
    x.tofile(tmp_path / "src_0.f32")
    r = subprocess.run([exe, "chain", str(tmp_path), str(ch), str(rate), str(block)] + ops, capture_output=True, text=True, timeout=300, env=dict(os.environ, RH_TEST_SOURCE=kind))
    what = (seed, (n, ch, rate, kind), ops, block)
    if r.returncode != 0:
        assert r.returncode == 1 and "unsupported" in r.stderr.lower(), (what, r.stderr)
        pytest.skip(f"refused: {r.stderr.strip()[:160]}")
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    ref = _oracle_full(O, M._span_source(O, kind, x, ch, rate), ops).collect()
    assert len(got) == len(ref), (what, len(got), len(ref))
    tol = _tolerance(ops, ref)
    if tol is None:
        assert np.array_equal(got, ref), (what, int(np.argmax(got != ref)))
    elif len(ref):
        if any(op.startswith("distortion") for op in ops):
            tol *= 8
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


@pytest.mark.parametrize("seed", list(range(60)))
def test_random_chain_of_any_adapters_over_spans_that_cut_frames(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _full_span_case(O, tmp_path, seed, FAKE)


# ------------------------------------------------------------------ ... and try_seek in the middle of a chain's stream ----
@pytest.mark.parametrize("seed", list(range(40)))
def test_random_seek_through_a_chain(O, tmp_path, seed):
    """try_seek through chains of seekable adapters whose state starts afresh behind a seek (amplify, filters, distortion, the limiter:
    blt.rs:350-377, limit.rs:1139-1158) over a SamplesBuffer, after any number of samples (also in the middle of a frame), to any position:
    what was pulled before the seek, then the chain applied afresh from the frame SamplesBuffer::try_seek lands on (buffer.rs:99-121: the float
    position, the next multiple of the channel count), resumed at the consumer's channel (tests/test_host_mirror.py::test_gpu_source_try_seek)."""
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    rng = np.random.default_rng(22000 + seed)
    ch, rate, n = int(rng.choice([1, 2, 2, 3, 6])), int(rng.choice(RATES)), int(rng.integers(100, 20000))
    x = M.rnd(22000 + seed, ch * n, 0.5)
    ops = []
    for _ in range(int(rng.integers(1, 4))):
        k = str(rng.choice(["amplify", "filter", "distortion", "limit"]))
        if k == "amplify":
            ops.append(f"amplify:{rng.choice([0.5, 0.8, 1.25])}")
        elif k == "filter":
            ops.append(f"{rng.choice(['low_pass', 'high_pass'])}:{rng.choice([800, 1000, 3000])}")
        elif k == "distortion":
            ops.append(f"distortion:{rng.choice([2.0, 4.0])}:{rng.choice([0.3, 0.6])}")
        elif "limit" not in ops:
            ops.append("limit")
    ops = ops or ["amplify:0.5"]
    block = int(rng.choice([64, 777, 4096, 16384]))
    pulled, seek_frame = int(rng.integers(0, ch * n)), int(rng.integers(0, n))
    ns = seek_frame * 1_000_000_000 // rate
    x.tofile(tmp_path / "src_0.f32")
    env = dict(os.environ, RH_TEST_SEEK_AFTER=str(pulled), RH_TEST_SEEK_NS=str(ns), RH_TEST_SOURCE="buffer")
    r = subprocess.run([FAKE, "chain", str(tmp_path), str(ch), str(rate), str(block)] + ops, capture_output=True, text=True, timeout=300, env=env)
    what = (seed, (n, ch, rate), ops, block, pulled, seek_frame)
    assert r.returncode == 0, (what, r.stderr)
    ok, k = (int(v) for v in (tmp_path / "seek.txt").read_text().split())
    assert ok == 1, what
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    secs = np.float32(ns // 1_000_000_000) + np.float32(ns % 1_000_000_000) / np.float32(1e9)  # math.rs:118-122
    npos = min(int(np.float32(np.float32(secs * np.float32(rate)) * np.float32(ch))), n * ch)
    frame = (npos + ch - 1) // ch
    before = _oracle_full(O, O.TestSource(x, ch, rate), ops).collect()[:k]
    after = _oracle_full(O, O.TestSource(x[frame * ch:], ch, rate), ops).collect()[k % ch:]
    ref = np.concatenate([before, after])
    assert len(got) == len(ref), (what, len(got), len(ref), k, frame)
    if len(ref):
        tol = 2 * TOL * max(1.0, float(np.max(np.abs(ref)))) * (8 if any(op.startswith("distortion") for op in ops) else 1)
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


# ------------------------------------------------------------------ ... and chains of any adapters handed to a mixer ----
def _mixer_chains_case(O, tmp_path, seed, exe):
    """`mixer.add(source.reverb(..).limit())` and the like: chains of the whole vocabulary (without dither; channel_volume's list of gains collides
    with the spec file's separator) over plain and spanned sources, handed over on the device or pulled through the host, into mixers of 1, 2
    and 6 channels -- against rodio's Mixer over UniformSourceIterator(chain.amplify(gain)).  Chains whose stream ends inside a frame (every
    second stereo reverb) included: GpuMixer completes them with their own `uniform`.  Combinations the mirror does not take are refused loudly."""
    rng = np.random.default_rng(33000 + seed)
    S, mixer_ch, to_rate = int(rng.integers(1, 5)), int(rng.choice([1, 2, 2, 6])), int(rng.choice([22050, 44100, 48000]))
    block, on_device = int(rng.choice([777, 4096, 20000])), bool(rng.integers(0, 2))
    kind = str(rng.choice(["test", "buffer", "mixed", "spans:2304", "spans:1000"]))
    lines, adds = [], []
    for i in range(S):
        gain = float(np.float32(rng.choice([0.5, 0.8, 1.0])))
        ch0, rate0 = int(rng.choice([1, 2, 2, 6])), int(rng.choice(RATES))
        x = M.rnd(33000 + 1000 * seed + 10 * i, int(rng.integers(1, 9000)) * ch0, 0.2)
        x.tofile(tmp_path / f"src_{i}.f32")
        ops = [o for o in _full_ops(rng, ch0, int(rng.integers(1, 3)), False) if not o.startswith("channel_volume")] if rng.random() < 0.6 else []
        lines.append(f"{ch0} {rate0} {gain} -1 0 {','.join(ops) if ops else '-'}\n")
        adds.append((x, ch0, rate0, i, ops, gain))
    (tmp_path / "spec.txt").write_text("".join(lines))
    r = subprocess.run([exe, "chainmix", str(tmp_path), str(S), str(mixer_ch), str(to_rate), str(block), "1" if on_device else "0"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RH_TEST_SOURCE=kind))
    what = (seed, S, mixer_ch, to_rate, block, on_device, kind, lines)
    if r.returncode != 0:
        assert r.returncode == 1 and "unsupported" in r.stderr.lower(), (what, r.stderr)
        pytest.skip(f"refused: {r.stderr.strip()[:160]}")
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    m = O.Mixer(mixer_ch, to_rate)
    for x, ch0, rate0, i, ops, gain in adds:
        m.add(O.UniformSourceIterator(_oracle_full(O, M._span_source(O, kind, x, ch0, rate0, i), ops).amplify(gain), mixer_ch, to_rate))
    ref = m.collect()
    assert len(got) == len(ref), (what, len(got), len(ref))
    if len(ref):
        tol = 2 * TOL * max(1.0, float(np.max(np.abs(ref)))) * (8 if any("agc" in l or "distortion" in l for l in lines) else 1)
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


@pytest.mark.parametrize("seed", list(range(60)))
def test_random_chains_of_any_adapters_into_a_mixer(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _mixer_chains_case(O, tmp_path, seed, FAKE)


# ------------------------------------------------------------------ ... and plain sources of any layout and span length into the fused stereo mixer ----
def _mixer_plain_case(O, tmp_path, seed, exe):
    """`mixer.add(source)` for one to eight plain sources of 1, 2, 3 or 6 channels at any rate into the stereo mixer, with and without the
    mixer's own filter: the fused path and its staging in front of it (one row per source, span by span).  The sources report no spans, one
    span (SamplesBuffer) or packets of 37, 1000, 2304, 32768 samples -- packets of 37 cut a stereo frame every time, and a fifth of the spanned
    sources END inside a frame.  Found here: a staging row filled to its last whole frame took a cut frame's samples into the next source's
    row (or past the block: seed 275, heap corruption)."""
    rng = np.random.default_rng(11000 + seed)
    S, to_rate = int(rng.integers(1, 9)), int(rng.choice([22050, 44100, 48000]))
    filt, freq = (int(rng.integers(0, 2)), int(rng.choice([200, 1000, 3000]))) if rng.random() < 0.5 else (-1, 0)
    block, lane = int(rng.choice([777, 4096, 16384, 30000])), int(rng.choice([3, 4, 8]))
    kind = str(rng.choice(["test", "buffer", "mixed", "spans:2304", "spans:1000", "spans:37", "spans:32768"]))
    kinds = [kind if kind != "mixed" else ["test", "buffer", "spans:4096"][i % 3] for i in range(S)]
    spec, xs = [], []
    for i in range(S):
        ch, rate, g = int(rng.choice([1, 2, 2, 2, 3, 6])), int(rng.choice(RATES)), float(np.float32(rng.choice([0.5, 0.8, 1.0, 1.2])))
        n = int(rng.integers(0, 20000)) * ch
        if kinds[i] != "test" and rng.random() < 0.2:  # (a source that reports no spans ends on a frame boundary: source/mod.rs:169-178)
            n += int(rng.integers(0, ch))
        spec.append((ch, rate, g, n))
        xs.append(M.rnd(11000 + 100 * seed + i, n, 0.15))
        xs[-1].tofile(tmp_path / f"src_{i}.f32")
    (tmp_path / "spec.txt").write_text("".join(f"{c} {r} {g}\n" for c, r, g, _ in spec))
    r = subprocess.run([exe, "mixany", str(tmp_path), str(S), str(to_rate), str(filt), str(freq), str(block), str(lane)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RH_TEST_SOURCE=kind))
    what = (seed, S, to_rate, filt, freq, block, lane, kind, spec)
    if r.returncode != 0:
        assert r.returncode == 1 and "unsupported" in r.stderr.lower(), (what, r.stderr)
        pytest.skip(f"refused: {r.stderr.strip()[:160]}")
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    ref = M._mix_oracle(O, spec, xs, kinds, 2, to_rate, filt, freq)
    assert len(got) == len(ref), (what, len(got), len(ref))
    if len(ref):
        tol = 2 * TOL * max(1.0, float(np.max(np.abs(ref))))
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


PLAIN_SEEDS = list(range(40)) + [45, 83, 275, 284]


@pytest.mark.parametrize("seed", PLAIN_SEEDS)
def test_random_plain_sources_into_the_stereo_mixer(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _mixer_plain_case(O, tmp_path, seed, FAKE)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", PLAIN_SEEDS[::4] + [275])
def test_gpu_random_plain_sources_into_the_stereo_mixer(O, tmp_path, seed):
    _mixer_plain_case(O, tmp_path, seed, M.EXE)


# ------------------------------------------------------------------ ... and sources whose spans cut frames into mixers of 1, 2, 3, 6 channels ----
def _mixer_cut_case(O, tmp_path, seed, exe):
    """`_mixer_case` where it hurts: sources of 1, 2, 3, 5, 6 channels as SamplesBuffers or in packets of 5, 37, 1000, 2304 samples (spans that end
    inside a frame all the time), three in ten ending inside a frame, with and without short adapter chains and per-source filters, into mixers
    of 1, 2, 3 and 6 channels.  Found here: a MONO mixer formed its mix in stereo and took channel 0 -- behind a span that gave the stereo
    stream an odd number of samples that was the wrong channel (a 7-channel SamplesBuffer; stereo packets of 37)."""
    rng = np.random.default_rng(77000 + seed)
    S, mixer_ch, to_rate = int(rng.integers(1, 6)), int(rng.choice([1, 2, 2, 6, 3])), int(rng.choice([22050, 44100, 48000]))
    block, on_device = int(rng.choice([777, 4096, 20000])), bool(rng.integers(0, 2))
    kind = str(rng.choice(["buffer", "mixed", "spans:2304", "spans:1000", "spans:37", "spans:5"]))
    kinds = [kind if kind != "mixed" else ["test", "buffer", "spans:4096"][i % 3] for i in range(S)]
    lines, adds = [], []
    for i in range(S):
        gain = float(np.float32(rng.choice([0.5, 0.8, 1.0, 1.2])))
        ch0, rate0 = int(rng.choice([1, 2, 2, 6, 3, 5])), int(rng.choice(RATES))
        n = int(rng.integers(1, 9000)) * ch0
        if kinds[i] != "test" and rng.random() < 0.3:
            n += int(rng.integers(0, ch0))
        x = M.rnd(77000 + 1000 * seed + 10 * i, n, 0.2)
        x.tofile(tmp_path / f"src_{i}.f32")
        ops = _ops_any_format(rng, int(rng.integers(1, 3)), [ch0]) if rng.random() < 0.4 else []
        fk, ff = (int(rng.integers(0, 2)), int(rng.choice([300, 1000, 3000]))) if (mixer_ch <= 2 and rng.random() < 0.4) else (-1, 0)
        lines.append(f"{ch0} {rate0} {gain} {fk} {ff} {','.join(ops) if ops else '-'}\n")
        adds.append((x, ch0, rate0, i, ops, gain, fk, ff))
    (tmp_path / "spec.txt").write_text("".join(lines))
    r = subprocess.run([exe, "chainmix", str(tmp_path), str(S), str(mixer_ch), str(to_rate), str(block), "1" if on_device else "0"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RH_TEST_SOURCE=kind))
    what = (seed, S, mixer_ch, to_rate, block, on_device, kind, lines)
    if r.returncode != 0:
        assert r.returncode == 1 and "unsupported" in r.stderr.lower(), (what, r.stderr)
        pytest.skip(f"refused: {r.stderr.strip()[:160]}")
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    m = O.Mixer(mixer_ch, to_rate)
    for x, ch0, rate0, i, ops, gain, fk, ff in adds:
        u = O.UniformSourceIterator(_oracle_chain(O, M._span_source(O, kind, x, ch0, rate0, i), ops).amplify(gain), mixer_ch, to_rate)
        m.add(u.low_pass(ff) if fk == 0 else u.high_pass(ff) if fk == 1 else u)
    ref = m.collect()
    assert len(got) == len(ref), (what, len(got), len(ref))
    if len(ref):
        tol = 2 * TOL * max(1.0, float(np.max(np.abs(ref)))) * (8 if any("agc" in l for l in lines) else 1)
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


CUT_SEEDS = list(range(40)) + [71, 110, 128, 130, 157, 171, 183]


@pytest.mark.parametrize("seed", CUT_SEEDS)
def test_random_sources_whose_spans_cut_frames_into_a_mixer(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _mixer_cut_case(O, tmp_path, seed, FAKE)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", CUT_SEEDS[::4] + [130])
def test_gpu_random_sources_whose_spans_cut_frames_into_a_mixer(O, tmp_path, seed):
    _mixer_cut_case(O, tmp_path, seed, M.EXE)


# ------------------------------------------------------------------ ... and late joins around mixes that end inside a frame ----
def _late_run(O, tmp_path, exe, spec, xs, kind, S0, mixer_ch, to_rate, block, pull_first):
    """`S0` sources, `pull_first` samples taken one by one, the rest of the sources added, everything else taken (the consumer keeps asking
    through up to 16 Nones, then until the first None): what rodio's mixer returns, what the mirror returns, and the Nones in between."""
    for i, x in enumerate(xs):
        x.tofile(tmp_path / f"src_{i}.f32")
    (tmp_path / "spec.txt").write_text("".join(f"{c} {r} {g}\n" for c, r, g in spec))
    chain = lambda i: O.UniformSourceIterator(M._span_source(O, kind, xs[i], spec[i][0], spec[i][1], i).amplify(spec[i][2]), mixer_ch, to_rate)
    m = O.Mixer(mixer_ch, to_rate)
    for i in range(S0):
        m.add(chain(i))
    ref = []
    for _ in range(pull_first):
        v = m.next()
        if v is None:
            break
        ref.append(v)
    for i in range(S0, len(xs)):
        m.add(chain(i))
    nones, v = 0, None
    while nones < 16:
        v = m.next()
        if v is not None:
            break
        nones += 1
    ref = np.concatenate([np.asarray(ref + [v], dtype=np.float32), m.collect()]) if v is not None else np.asarray(ref, dtype=np.float32)
    r = subprocess.run([exe, "latewide", str(tmp_path), str(S0), str(len(xs) - S0), str(mixer_ch), str(to_rate), str(block), str(pull_first)], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, RH_TEST_SOURCE=kind))
    if r.returncode != 0:
        assert r.returncode == 1 and "unsupported" in r.stderr.lower(), r.stderr
        pytest.skip(f"refused: {r.stderr.strip()[:160]}")
    return np.fromfile(tmp_path / "out.f32", dtype=np.float32), ref, int((tmp_path / "nones.txt").read_text()), nones


def _late_cut_case(O, tmp_path, seed, exe):
    """`_late_case` over sources whose spans cut frames (SamplesBuffers of 3, 5, 6 channels, packets of 5, 37, 1000, 2304 samples), three in ten
    ending inside a frame, into mixers of 1, 2, 3 and 6 channels.  Found here: a mix that was about to END inside a frame -- its last block
    already cut to the samples rodio's sources cover, one block ahead of the consumer -- stayed cut when a source joined in front of that
    frame (a sample of the newcomer missing, up to five in a 6-channel mixer); and a newcomer that had two scheduled blocks of a generation
    behind a full queue to catch up with overran its own queue."""
    rng = np.random.default_rng(56000 + seed)
    S0, S1 = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    mixer_ch, to_rate, block = int(rng.choice([1, 2, 2, 6, 3])), int(rng.choice([22050, 44100, 48000])), int(rng.choice([777, 4096, 7000]))
    kind = str(rng.choice(["buffer", "mixed", "spans:2304", "spans:37", "spans:1000", "spans:5"]))
    kinds = [kind if kind != "mixed" else ["test", "buffer", "spans:4096"][i % 3] for i in range(S0 + S1)]
    spec, xs = [], []
    for i in range(S0 + S1):
        ch, rate, gain = int(rng.choice([1, 2, 2, 6, 3, 5])), int(rng.choice(RATES)), float(np.float32(rng.choice([0.5, 0.8, 1.0])))
        n = int(rng.integers(1, 12000)) * ch
        if kinds[i] != "test" and rng.random() < 0.3:
            n += int(rng.integers(0, ch))
        spec.append((ch, rate, gain))
        xs.append(M.rnd(56000 + 100 * seed + i, n, 0.2))
    total0 = len(O_collect_copy(O, kind, xs, spec, S0, mixer_ch, to_rate))
    pull_first = int(rng.integers(0, max(1, int(total0 * 1.2)) + 1))
    got, ref, nones_got, nones = _late_run(O, tmp_path, exe, spec, xs, kind, S0, mixer_ch, to_rate, block, pull_first)
    what = (seed, S0, S1, mixer_ch, to_rate, block, kind, pull_first, total0, spec, [len(x) for x in xs])
    assert len(got) == len(ref), (what, len(got), len(ref))
    assert float(np.max(np.abs(got - ref))) <= TOL if len(ref) else True, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))
    assert nones_got == nones, what


LATE_CUT_SEEDS = list(range(40)) + [50, 54, 58, 78, 88, 89, 109, 136, 161, 169, 191, 194]


@pytest.mark.parametrize("seed", LATE_CUT_SEEDS)
def test_random_late_joins_of_sources_whose_spans_cut_frames(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _late_cut_case(O, tmp_path, seed, FAKE)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", LATE_CUT_SEEDS[::4] + [161, 194])
def test_gpu_random_late_joins_of_sources_whose_spans_cut_frames(O, tmp_path, seed):
    _late_cut_case(O, tmp_path, seed, M.EXE)


LATE_EDGES = [  # (specs, samples, mixer channels, pull_first): a first source that ends inside a frame, a second one added around that frame
    ([(2, 48000, 1.0), (2, 48000, 0.5)], [2001, 3000], 2, pf) for pf in (1990, 1999, 2000, 2001, 2002)
] + [([(6, 48000, 1.0), (6, 48000, 0.5)], [603, 3000], 6, pf) for pf in (600, 601, 602, 603, 604, 605, 606)] + [([(6, 48000, 1.0), (2, 44100, 0.5)], [603, 300], 6, pf) for pf in (601, 603)]


def _late_edge(O, tmp_path, exe, case):
    spec, ns, mixer_ch, pull_first = case
    xs = [M.rnd(900 + i, n, 0.2) for i, n in enumerate(ns)]
    got, ref, nones_got, nones = _late_run(O, tmp_path, exe, spec, xs, "buffer", 1, mixer_ch, 48000, 777, pull_first)
    assert len(got) == len(ref), (case, len(got), len(ref))
    assert np.array_equal(got, ref), (case, int(np.argmax(got != ref)))
    assert nones_got == nones, (case, nones_got, nones)


@pytest.mark.parametrize("case", LATE_EDGES)
def test_late_join_around_a_mix_that_ends_inside_a_frame(O, tmp_path, case):
    """The first source's stream ends inside a frame, and the consumer is one block behind the mixer, which has cut its last block to rodio's
    samples.  A source that joins IN FRONT of that frame completes it (the block gets its whole length back); one that joins while the
    consumer is INSIDE it waits: rodio's mixer has no source left for the rest of the frame -- None, once per missing sample -- and starts
    the pending one at the next frame (mixer.rs:120-136,175-183).  Bit for bit, and the same number of Nones."""
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _late_edge(O, tmp_path, FAKE, case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", LATE_EDGES[::2])
def test_gpu_late_join_around_a_mix_that_ends_inside_a_frame(O, tmp_path, case):
    _late_edge(O, tmp_path, M.EXE, case)


# ------------------------------------------------------------------ ... and the span arithmetic of take_duration / delay in front of an iterator that asks ----
def _span_arithmetic_case(O, tmp_path, seed, exe):
    """`TakeDuration::current_span_len()` answers Some(what the duration still admits) -- over a generator that says None too, and Some(0) once
    it is spent (take.rs:176-195); `Delay` adds the silence it still owes (delay.rs:94-98).  A UniformSourceIterator behind them (the chain's own
    `.uniform()`, or the mixer's) therefore converts in chains of 32768 samples where it would convert a bare generator in one, and ends in
    front of the silence that completes a frame the duration cut.  Continuous sources of up to 60 000 frames, takes that end before and behind
    the source's end, inside and between frames; as a chain of its own (half of them) and into mixers of 1, 2, 6 channels.  (Until round 5
    the oracle handed the input's answer through -- as did the mirror.  Both follow take.rs now.)"""
    rng = np.random.default_rng(88000 + seed)
    ch, rate = int(rng.choice([1, 2, 2, 3, 6])), int(rng.choice(RATES))
    frames = int(rng.integers(100, 60000))
    # ... and over sources that report spans themselves: the adapters ask THEM wherever the consumer asks the adapters (`min(the input's span,
    # what the duration admits)`, `the input's span + the silence owed`), which the chain answers from the upstream's answers by sample position
    kind = str(rng.choice(["test", "test", "buffer", "spans:1000", "spans:37", "spans:40000", "spans:5"]))
    # (a SamplesBuffer may end inside a frame; the packet sources of this harness keep answering Some(packet) when they are exhausted, and what
    # rodio's ChannelVolume returns when it is asked AGAIN after the None of a cut frame -- the stale sum, channel_volume.rs:71-88 -- is theirs alone)
    x = M.rnd(88000 + seed, frames * ch + (int(rng.integers(0, ch)) if kind == "buffer" and rng.random() < 0.3 else 0), 0.4)
    x.tofile(tmp_path / "src_0.f32")
    env = dict(os.environ, RH_TEST_SOURCE=kind)
    in_chain = rng.random() < 0.5
    dur = lambda f: int(f * 1e9 / rate) + int(rng.integers(0, 30000))  # nanoseconds for about f frames
    ops = []
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(0, 6))
        if k == 0 or not ops:
            ops.append(f"take:{dur(rng.integers(1, int(frames * 1.3) + 2))}:{int(rng.integers(0, 2))}")
        elif k == 1:
            ops.append(f"delay:{dur(rng.integers(0, 3000))}")
        elif k == 2:
            ops.append(f"amplify:{float(np.float32(rng.choice([0.5, 0.8, 1.2])))}")
        elif k == 3:
            ops.append(f"low_pass:{int(rng.choice([300, 1000, 3000]))}")
        elif k == 4:
            ops.append("limit")
        else:
            ops.append(f"fade_in:{dur(rng.integers(1, 2000))}")
    if in_chain and rng.random() < 0.3:  # ChannelVolume hands its input's answer on although it changes the sample count (channel_volume.rs:103-105)
        ops.insert(int(rng.integers(0, len(ops) + 1)), "channel_volume:" + ",".join(str(float(np.float32(g))) for g in rng.choice([0.25, 0.5, 1.0], int(rng.integers(1, 4)))))
    block = int(rng.choice([777, 4096, 40000]))
    if in_chain:  # the chain's own iterator
        ops = ops + [f"uniform:{int(rng.choice([1, 2, 6]))}:{int(rng.choice([22050, 44100, 48000]))}"] + (["amplify:0.5"] if rng.random() < 0.5 else [])
        r = subprocess.run([exe, "chain", str(tmp_path), str(ch), str(rate), str(block)] + ops, capture_output=True, text=True, timeout=300, env=env)
        what = (seed, (frames, ch, rate), kind, ops, block)
        ref = _oracle_full(O, M._span_source(O, kind, x, ch, rate, 0), ops).collect()
    else:  # the mixer's
        mixer_ch, to_rate, on_device = int(rng.choice([1, 2, 2, 6])), int(rng.choice([22050, 44100, 48000])), bool(rng.integers(0, 2))
        gain = float(np.float32(rng.choice([0.5, 1.0])))
        (tmp_path / "spec.txt").write_text(f"{ch} {rate} {gain} -1 0 {','.join(ops)}\n2 44100 0.5 -1 0 -\n")
        y = M.rnd(88500 + seed, 2 * 3000, 0.2)
        y.tofile(tmp_path / "src_1.f32")
        r = subprocess.run([exe, "chainmix", str(tmp_path), "2", str(mixer_ch), str(to_rate), str(block), "1" if on_device else "0"], capture_output=True, text=True, timeout=300, env=env)
        what = (seed, (frames, ch, rate), kind, ops, block, mixer_ch, to_rate, on_device)
        m = O.Mixer(mixer_ch, to_rate)
        m.add(O.UniformSourceIterator(_oracle_full(O, M._span_source(O, kind, x, ch, rate, 0), ops).amplify(gain), mixer_ch, to_rate))
        m.add(O.UniformSourceIterator(M._span_source(O, kind, y, 2, 44100, 1).amplify(0.5), mixer_ch, to_rate))
        ref = m.collect()
    if r.returncode != 0:
        assert r.returncode == 1 and "unsupported" in r.stderr.lower(), (what, r.stderr)
        pytest.skip(f"refused: {r.stderr.strip()[:160]}")
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert len(got) == len(ref), (what, len(got), len(ref))
    if len(ref):
        nan = np.isnan(ref)  # (a fade-out over less than a millisecond is 0 / 0 in rodio: take.rs:33-38)
        assert np.array_equal(np.isnan(got), nan), what
        got, ref = got[~nan], ref[~nan]
    if len(ref):
        tol = 2 * TOL * max(1.0, float(np.max(np.abs(ref))))
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))


@pytest.mark.parametrize("seed", list(range(60)))
def test_random_takes_and_delays_in_front_of_an_iterator_that_asks_for_spans(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _span_arithmetic_case(O, tmp_path, seed, FAKE)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(0, 60, 3)))
def test_gpu_random_takes_and_delays_in_front_of_an_iterator_that_asks_for_spans(O, tmp_path, seed):
    _span_arithmetic_case(O, tmp_path, seed, M.EXE)


# ------------------------------------------------------------------ ... and chains and filtered sources added to a RUNNING mixer ----
def _late_chains_case(O, tmp_path, seed, exe):
    """`_mixer_chains_case` + `_late_case`: one to three sources -- plain, with a filter of their own, or chains of the whole vocabulary, handed over
    on the device or through the host -- play in a mixer of 1, 2 or 6 channels; after `pull_first` samples one or two more are added
    (mixer.rs:175-183: admitted at the next frame), sources of any span kind.  Samples and the Nones in between as rodio's mixer returns them."""
    rng = np.random.default_rng(58000 + seed)
    S0, S1 = int(rng.integers(1, 4)), int(rng.integers(1, 3))
    mixer_ch, to_rate = int(rng.choice([1, 2, 2, 6])), int(rng.choice([22050, 44100, 48000]))
    block, on_device = int(rng.choice([777, 4096, 7000])), bool(rng.integers(0, 2))
    kind = str(rng.choice(["test", "buffer", "mixed", "spans:1000", "spans:37"]))
    lines, adds = [], []
    for i in range(S0 + S1):
        ch0, rate0, gain = int(rng.choice([1, 2, 2, 6])), int(rng.choice(RATES)), float(np.float32(rng.choice([0.5, 0.8, 1.0])))
        x = M.rnd(58000 + 100 * seed + i, int(rng.integers(1, 9000)) * ch0, 0.2)
        x.tofile(tmp_path / f"src_{i}.f32")
        ops = [o for o in _full_ops(rng, ch0, int(rng.integers(1, 3)), False) if not o.startswith("channel_volume")] if rng.random() < 0.6 else []
        fk, ff = (int(rng.integers(0, 2)), int(rng.choice([300, 1000, 3000]))) if (mixer_ch <= 2 and rng.random() < 0.3) else (-1, 0)
        lines.append(f"{ch0} {rate0} {gain} {fk} {ff} {','.join(ops) if ops else '-'}\n")
        adds.append((x, ch0, rate0, i, ops, gain, fk, ff))
    (tmp_path / "spec.txt").write_text("".join(lines))

    def chain(a):
        x, ch0, rate0, i, ops, gain, fk, ff = a
        u = O.UniformSourceIterator(_oracle_full(O, M._span_source(O, kind, x, ch0, rate0, i), ops).amplify(gain), mixer_ch, to_rate)
        return u.low_pass(ff) if fk == 0 else u.high_pass(ff) if fk == 1 else u

    m0 = O.Mixer(mixer_ch, to_rate)
    for a in adds[:S0]:
        m0.add(chain(a))
    total0 = len(m0.collect())
    pull_first = int(rng.integers(0, max(1, int(total0 * 1.2)) + 1))
    m = O.Mixer(mixer_ch, to_rate)
    for a in adds[:S0]:
        m.add(chain(a))
    ref = []
    for _ in range(pull_first):
        v = m.next()
        if v is None:
            break
        ref.append(v)
    for a in adds[S0:]:
        m.add(chain(a))
    nones, v = 0, None
    while nones < 16:
        v = m.next()
        if v is not None:
            break
        nones += 1
    ref = np.concatenate([np.asarray(ref + [v], dtype=np.float32), m.collect()]) if v is not None else np.asarray(ref, dtype=np.float32)
    r = subprocess.run([exe, "latechain", str(tmp_path), str(S0), str(S1), str(mixer_ch), str(to_rate), str(block), "1" if on_device else "0", str(pull_first)], capture_output=True,
                       text=True, timeout=300, env=dict(os.environ, RH_TEST_SOURCE=kind))
    what = (seed, S0, S1, mixer_ch, to_rate, block, on_device, kind, pull_first, total0, lines)
    if r.returncode != 0:
        assert r.returncode == 1 and "unsupported" in r.stderr.lower(), (what, r.stderr)
        pytest.skip(f"refused: {r.stderr.strip()[:160]}")
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert len(got) == len(ref), (what, len(got), len(ref))
    if len(ref):
        nan = np.isnan(ref)  # (a fade-out over less than a millisecond is 0 / 0 in rodio: take.rs:33-38)
        assert np.array_equal(np.isnan(got), nan), what
        got, ref = got[~nan], ref[~nan]
    if len(ref):
        tol = 2 * TOL * max(1.0, float(np.max(np.abs(ref)))) * (8 if any("agc" in l or "distortion" in l for l in lines) else 1)
        assert float(np.max(np.abs(got - ref))) <= tol, (what, float(np.max(np.abs(got - ref))), int(np.argmax(np.abs(got - ref))))
    assert int((tmp_path / "nones.txt").read_text()) == nones, what


@pytest.mark.parametrize("seed", list(range(48)) + [611])
def test_random_chains_and_filtered_sources_join_a_running_mixer(O, tmp_path, seed):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    _late_chains_case(O, tmp_path, seed, FAKE)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(0, 48, 4)) + [611])
def test_gpu_random_chains_and_filtered_sources_join_a_running_mixer(O, tmp_path, seed):
    _late_chains_case(O, tmp_path, seed, M.EXE)
