"""The HOST LOGIC of include/rodio_hip.hpp without a GPU: tests/cpp/host_mirror_test linked against tests/cpp/fake_device.cpp (a CPU stand-in
for librodio_hip.so -- test infrastructure, see its header) instead of the library, so that span readers, planners, block pumps, generations,
late joins and format changes run in the `-m "not gpu"` suite.  The cases are the ones of tests/test_host_mirror.py (which run them on the GPU,
through the real library): the same driver arguments, the same oracle chains, the same tolerances -- here they check the host side only."""
import itertools
import os
import subprocess

import numpy as np
import pytest

import test_host_mirror as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "cpp", "host_mirror_test_fake")


@pytest.fixture()
def fake(monkeypatch):
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    monkeypatch.setattr(M, "EXE", FAKE)
    return FAKE


def test_the_fake_device_is_linked_into_the_test_driver_only():
    # the product never sees it: no file under rodio_amd/, include/, rust/ or the repo root mentions it, and librodio_hip.so does not export its marker
    for top in ("rodio_amd", "include", "rust"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".h", ".hpp", ".hip", ".rs", ".toml")) and not (top == "rodio_amd" and f == "build.py"):  # (build.py compiles the test drivers)
                    assert "fake_device" not in open(os.path.join(dirpath, f), errors="replace").read(), os.path.join(dirpath, f)
    for f in ("bench.py", "__graft_entry__.py"):
        assert "fake_device" not in open(os.path.join(ROOT, f)).read() and "host_mirror_test_fake" not in open(os.path.join(ROOT, f)).read()
    nm = subprocess.run(["nm", "-D", "--undefined-only", FAKE], capture_output=True, text=True).stdout
    assert not [l for l in nm.splitlines() if " rh_" in l], "the fake driver must not pull entry points from librodio_hip.so"


def test_selftest(fake):
    r = subprocess.run([fake, "selftest"], capture_output=True, text=True)
    assert r.returncode == 0 and "selftest ok" in r.stdout, r.stderr


@pytest.mark.parametrize("filt,freq,block,R", [(-1, 0, 4096, 4), (0, 200, 30000, 8), (1, 300, 1000, 3)])
def test_mixer_pull(O, tmp_path, fake, filt, freq, block, R):
    M.test_gpu_mixer_pull_equals_rodio_chain(O, tmp_path, filt, freq, block, R)


@pytest.mark.parametrize("filt,freq", [(-1, 0), (0, 300)])
def test_any_source_layout(O, tmp_path, fake, filt, freq):
    M.test_gpu_mixer_takes_any_source_layout(O, tmp_path, filt, freq)


@pytest.mark.parametrize("filt,freq,pull_first", [(-1, 0, 11), (0, 200, 8704 * 2 + 1), (-1, 0, 200000), (0, 200, 0)])
def test_add_on_a_running_mixer(O, tmp_path, fake, filt, freq, pull_first):
    M.test_gpu_mixer_add_on_a_running_mixer(O, tmp_path, filt, freq, pull_first)


@pytest.mark.parametrize("kind,filt,freq,block", [("buffer", -1, 0, 777), ("spans:2304", 0, 200, 16384), ("mixed", -1, 0, 16384)])
def test_spanned_sources(O, tmp_path, fake, kind, filt, freq, block):
    M.test_gpu_mixer_converts_spanned_sources_span_by_span(O, tmp_path, kind, filt, freq, block)


@pytest.mark.parametrize("kind,filt,freq,pull_first", [("buffer", -1, 0, 11), ("spans:1000", 0, 200, 40000)])
def test_late_join_of_spanned_sources(O, tmp_path, fake, kind, filt, freq, pull_first):
    M.test_gpu_mixer_late_join_of_spanned_sources(O, tmp_path, kind, filt, freq, pull_first)


@pytest.mark.parametrize("case,kind,block", [(0, "buffer", 777), (1, "spans:1500", 16384), (3, "spans:32768", 777), (4, "test", 16384)])
def test_source_uniform(O, tmp_path, fake, case, kind, block):
    M.test_gpu_source_uniform_converts_span_by_span(O, tmp_path, case, kind, block)


@pytest.mark.parametrize("block", [4096, 20000])
def test_a_filter_per_source(O, tmp_path, fake, block):
    M.test_gpu_mixer_a_filter_per_source(O, tmp_path, block)


@pytest.mark.parametrize("mixer_ch", [1, 2])
def test_output_layouts(O, tmp_path, fake, mixer_ch):
    M.test_gpu_mixer_output_layouts(O, tmp_path, mixer_ch)


# ---- round 5: what was refused until now ----
@pytest.mark.parametrize("filt,freq,ch,samples,rate", [(-1, 0, 6, 100000, 44100), (0, 200, 6, 100000, 44100), (-1, 0, 3, 70002, 44100), (-1, 0, 5, 99999, 44100), (-1, 0, 7, 100000, 44100),
                                                       (0, 200, 7, 65537, 48000), (-1, 0, 5, 40003, 8000), (-1, 0, 6, 100000, 96000)])
def test_a_cut_frame_in_front_of_a_rate_conversion(O, tmp_path, fake, filt, freq, ch, samples, rate):
    M.test_gpu_mixer_a_cut_frame_in_front_of_a_rate_conversion(O, tmp_path, filt, freq, ch, samples, rate)


@pytest.mark.parametrize("ch,samples,rate", [(6, 100000, 44100), (7, 65537, 48000), (5, 33000, 44100)])
def test_a_mix_that_ends_inside_a_frame(O, tmp_path, fake, ch, samples, rate):
    M.test_gpu_mixer_a_mix_that_ends_inside_a_frame(O, tmp_path, ch, samples, rate)


@pytest.mark.parametrize("mixer_ch,block,kind", [(6, 4096, "test"), (6, 20000, "buffer"), (4, 4096, "mixed"), (3, 20000, "buffer")])
def test_mixers_of_more_than_two_channels(O, tmp_path, fake, mixer_ch, block, kind):
    M.test_gpu_mixer_of_more_than_two_channels(O, tmp_path, mixer_ch, block, kind)


@pytest.mark.parametrize("mixer_ch,block", [(6, 4096), (8, 1000), (3, 65536)])
def test_wide_generation_in_one_launch_a_block(O, tmp_path, fake, mixer_ch, block):
    M.test_gpu_mixer_wide_generation_in_one_launch_a_block(O, tmp_path, mixer_ch, block)


def test_wide_generation_with_one_chain_among_the_plain_sources(O, tmp_path, fake):
    M.test_gpu_mixer_wide_generation_with_one_chain_among_the_plain_sources(O, tmp_path)


@pytest.mark.parametrize("on_device", [True, False])
def test_six_channels_filters_and_chains(O, tmp_path, fake, on_device):
    M.test_gpu_mixer_of_six_channels_filters_and_chains(O, tmp_path, on_device)


@pytest.mark.parametrize("pull_first", [11, 6 * 9000 + 1])
def test_six_channels_late_join(O, tmp_path, fake, pull_first):
    M.test_gpu_mixer_of_six_channels_late_join(O, tmp_path, pull_first)


@pytest.mark.parametrize("case,kind,block", [(c, k, b) for (c, k), b in zip(itertools.product(range(len(M.CUT_CHAINS)), ["buffer", "spans:37"]), itertools.cycle([777, 16384]))])
def test_source_uniform_spans_that_cut_a_frame(O, tmp_path, fake, case, kind, block):
    M.test_gpu_source_uniform_spans_that_cut_a_frame(O, tmp_path, case, kind, block)


@pytest.mark.parametrize("case,block", [(c, b) for c, b in zip(range(len(M.SEQ_CHAINS)), itertools.cycle([777, 16384]))])
def test_source_follows_a_format_change(O, tmp_path, fake, case, block):
    M.test_gpu_source_follows_a_format_change(O, tmp_path, case, block)


def test_refusals_across_a_format_change(tmp_path, fake):
    M.test_gpu_source_refuses_what_it_does_not_mirror_across_a_format_change(tmp_path)


@pytest.mark.parametrize("mixer_ch,block", [(2, 4096), (6, 20000)])
def test_mixer_takes_sources_that_change_their_format(O, tmp_path, fake, mixer_ch, block):
    M.test_gpu_mixer_takes_sources_that_change_their_format(O, tmp_path, mixer_ch, block)


def test_every_case_of_the_gpu_suite_of_the_host_mirror_on_the_stand_in_device():
    """tests/test_host_mirror.py's `-m gpu` cases -- all 266 of them, not the selection above -- with the driver that is linked against
    tests/cpp/fake_device.cpp (RH_HOST_MIRROR_EXE): the host logic they exercise runs here without a GPU (the stand-in has every entry point
    the mirror calls; `rh_dither` with the counter-based noise rodio_hip.h states for it)."""
    assert os.path.exists(FAKE), "run python rodio_amd/build.py"
    import sys

    cmd = [sys.executable, "-m", "pytest", "tests/test_host_mirror.py", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-x"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200, env=dict(os.environ, RH_HOST_MIRROR_EXE=FAKE))
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, r.stdout[-3000:]
    assert int(tail.split(" passed")[0].split()[-1]) >= 266, tail
