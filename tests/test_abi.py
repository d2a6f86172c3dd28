"""The C-ABI library loads on a CPU-only box and exports every symbol include/rodio_hip.h
declares; without a GPU it refuses to compute instead of falling back.  No kernels run here."""
import ctypes as C
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rodio_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(rh):
    lib = C.CDLL(rh.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_header(rh):
    from rodio_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared()


def test_refuses_to_compute_without_gpu(rh):
    import torch

    if torch.cuda.is_available():
        return  # covered by the gpu tests
    from rodio_amd import _lib

    lib = _lib.lib
    assert lib.rh_init(0) == 2  # RH_ERR_HIP: no device, and no CPU fallback
    assert b"device" in lib.rh_last_hip_error().lower()
    buf = (C.c_float * 4)()
    # every compute entry point refuses: RH_ERR_NOT_INITIALIZED (6)
    assert lib.rh_amplify(buf, buf, 4, 2.0, None) == 6
    assert lib.rh_resample_linear(buf, buf, 2, 44100, 48000, 1, 0, None) == 6
    h = C.c_void_p()
    cfg = _lib.RlmConfig(44100, 48000, 2, 0, 0, 200, 0.5, 4, 1024, 0, 0)
    assert lib.rh_rlm_create(C.byref(h), C.byref(cfg)) == 6


def test_status_strings(rh):
    from rodio_amd import _lib

    for code in range(8):
        assert _lib.lib.rh_status_string(code)
    assert _lib.lib.rh_version() >= 100


def test_headers_compile_cleanly(tmp_path):
    """include/rodio_hip.h is plain C (C99, what an FFI generator reads); include/rodio_hip.hpp is C++17 over nothing but
    that header.  Both without a warning under -Wall -Wextra -Wpedantic."""
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    if not (shutil.which("gcc") and shutil.which("g++")):
        import pytest

        pytest.skip("no host compiler")
    c = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Wpedantic", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, "rodio_hip.h")], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr
    src = tmp_path / "hdr.cpp"
    src.write_text('#include "rodio_hip.hpp"\nint main() { return 0; }\n')
    cxx = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Wpedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], capture_output=True, text=True)
    assert cxx.returncode == 0, cxx.stderr
