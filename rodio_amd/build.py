"""Builds librodio_hip.so (gfx950 only) in-tree with hipcc.

    python rodio_amd/build.py [--force]      (run by path: importing the package needs the .so)

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with
the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "librodio_hip.so")
SOURCES = ["rh_runtime.hip", "rh_elementwise.hip", "rh_resample.hip", "rh_recurrence.hip", "rh_limit.hip", "rh_agc.hip", "rh_biquad_scan.hip", "rh_stream.hip", "rh_uniform.hip", "rh_widemix.hip", "rh_formats.hip", "rh_wav.hip", "rh_comm.hip", "rh_pipeline.hip", "rh_pipeline_plan.hip", "rh_pipeline_stream.hip", "rh_pipeline_sblk.hip"]
# -ffp-contract=off: the reference's f32 expressions (lerp, biquad, mixer sum) must not be
# fused; kernels that want an FMA spell it __builtin_fmaf.
# -fno-slp-vectorize: hipcc's SLP pass pairs the two stereo channels into v_pk_*_f32; on gfx950
# that costs more v_mov shuffling and VGPRs than it saves (measured: fused kernel 1.25 -> 0.89 ms).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"] + os.environ.get("RH_EXTRA_HIPCC_FLAGS", "").split()


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: librodio_hip.so cannot be built (there is no CPU fallback)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, "rh_common.h"), os.path.join(CSRC, "rh_scan_common.h"), os.path.join(CSRC, "rh_pipeline_internal.h"), os.path.join(CSRC, "rh_pipeline_dev.h"), os.path.join(HERE, "..", "include", "rodio_hip.h")]
    cc = hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([cc, *FLAGS, "-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"])
    build_host_mirror_test(force, run)
    return LIB


def build_host_mirror_test(force: bool, run) -> str:
    """The C++ host mirror (include/rodio_hip.hpp) is header-only; its test driver is a plain g++ program over
    the C ABI -- no HIP headers, the way a host application links the library."""
    root = os.path.join(HERE, "..")
    src = os.path.join(root, "tests", "cpp", "host_mirror_test.cpp")
    exe = os.path.join(root, "tests", "cpp", "host_mirror_test")
    deps = [src, os.path.join(root, "include", "rodio_hip.hpp"), os.path.join(root, "include", "rodio_hip.h"), LIB]
    if force or _stale(exe, deps):
        run([shutil.which("g++") or "g++", "-std=c++17", "-O2", "-pthread", "-Wall", "-Wextra", "-I", os.path.join(root, "include"), src, "-L", HERE, "-lrodio_hip",
             "-Wl,-rpath,$ORIGIN/../../rodio_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe])
    # ... and the same driver over tests/cpp/fake_device.cpp (a CPU stand-in for the library: TEST INFRASTRUCTURE, it lets the host logic of the
    # header run in the `-m "not gpu"` suite; nothing of the product links or loads it)
    fake_src = os.path.join(root, "tests", "cpp", "fake_device.cpp")
    fake_exe = os.path.join(root, "tests", "cpp", "host_mirror_test_fake")
    if force or _stale(fake_exe, [src, fake_src, deps[1], deps[2]]):
        run([shutil.which("g++") or "g++", "-std=c++17", "-O2", "-ffp-contract=off", "-pthread", "-Wall", "-Wextra", "-I", os.path.join(root, "include"), src, fake_src, "-o", fake_exe])
    return exe


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
