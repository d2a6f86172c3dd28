import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The suite needs what `__graft_entry__.build()` makes (the library, the oracle's .so, the two test drivers): they are git-ignored, so a
    fresh checkout has none of them.  Build once if any is missing (hipcc cross-compiles without a GPU); a failure here shows up as the tests'
    own "run python rodio_amd/build.py"."""
    need = [os.path.join(ROOT, "rodio_amd", "librodio_hip.so"), os.path.join(ROOT, "oracle", "librodio_oracle.so"),
            os.path.join(ROOT, "tests", "cpp", "host_mirror_test"), os.path.join(ROOT, "tests", "cpp", "host_mirror_test_fake")]
    if all(os.path.exists(p) for p in need):
        return
    import subprocess

    # (a failed build used to surface as hundreds of "run python rodio_amd/build.py" failures with the compiler's output thrown away)
    try:
        r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, timeout=3000, check=False, capture_output=True, text=True)
    except Exception as e:  # noqa: BLE001
        pytest.exit(f"building the library / oracle / test drivers failed: {e}", returncode=3)
    if r.returncode != 0:
        pytest.exit("building the library / oracle / test drivers failed (python -c 'import __graft_entry__ as g; g.build()'):\n" + (r.stderr or r.stdout)[-4000:], returncode=3)


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (oracle/rodio_oracle.py).  Test infrastructure only."""
    from oracle import rodio_oracle

    return rodio_oracle


@pytest.fixture(scope="session")
def rh():
    """The product package, with the HIP library loaded.  Fails loudly if the .so is missing."""
    import rodio_amd

    return rodio_amd


def sine_generator(sample_rate: int, frequency: float, n: int):
    """Test input only: SignalGenerator::new(rate, f, Sine) restated (src/source/signal_generator.rs:51-53,107-135):
    period = rate / f, phase_step = 1 / period, sample = sin(TAU * phase), phase = (phase + step).rem_euclid(1), all f32."""
    import numpy as np

    f32 = np.float32
    step = f32(1.0) / (f32(sample_rate) / f32(frequency))
    phase = f32(0.0)
    ph = np.empty(n, dtype=np.float32)
    for i in range(n):
        ph[i] = phase
        phase = f32(phase + step)
        phase = f32(phase - np.floor(phase))  # rem_euclid(1.0) of a non-negative value
    return np.sin(f32(6.2831855) * ph).astype(np.float32)


import contextlib


@contextlib.contextmanager
def knobs(**env):
    """The library's diagnostic / tuning variables are read once, by rh_init(): set them, re-read, run, restore."""
    import rodio_amd

    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    rodio_amd.init(0)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        rodio_amd.init(0)
