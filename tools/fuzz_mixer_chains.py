"""More seeds of tests/test_host_logic_fuzz_cpu.py's random cases than the suite runs (the host mirror on the CPU stand-in of the C ABI against the
oracle): chains handed to a mixer by default, any of the other families by name.

    python tools/fuzz_mixer_chains.py [first_seed last_seed [family]]      family: mixer_chains (default) | sequence | one_source | full | full_span | mixer | late | mixer_plain | mixer_cut | late_cut | span_arithmetic | late_chains

RH_FUZZ_EXE=<binary> runs another build of the driver -- the one with the address sanitizer that found the staging row overflow:
    g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -ffp-contract=off -pthread -I include tests/cpp/host_mirror_test.cpp tests/cpp/fake_device.cpp -o /tmp/hmt_asan
"""
import os
import pathlib
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import pytest  # noqa: E402
import test_host_logic_fuzz_cpu as F  # noqa: E402
from oracle import rodio_oracle as O  # noqa: E402

lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 200)
family = sys.argv[3] if len(sys.argv) > 3 else "mixer_chains"
fn = {"mixer_chains": F._mixer_chains_case, "sequence": F._sequence_case, "one_source": F._one_source_case, "full": lambda o, t, s, e: F._full_case(o, t, s, e, dither=False),
      "full_span": F._full_span_case, "mixer": F._mixer_case, "late": F._late_case, "mixer_plain": F._mixer_plain_case, "mixer_cut": F._mixer_cut_case, "late_cut": F._late_cut_case, "span_arithmetic": F._span_arithmetic_case, "late_chains": F._late_chains_case}[family]
bad = refused = 0
for seed in range(lo, hi):
    tmp = pathlib.Path(tempfile.mkdtemp())
    try:
        fn(O, tmp, seed, os.environ.get("RH_FUZZ_EXE", F.FAKE))
    except pytest.skip.Exception:
        refused += 1
    except (AssertionError, Exception) as e:
        bad += 1
        print(family, seed, str(e)[:600])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
print("bad", bad, "refused", refused)
