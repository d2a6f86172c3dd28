"""Instruction mix of one kernel from hipcc's assembly (-S): how many VALU / SALU / LDS / VMEM instructions the code holds, and how many of the
VALU ones only move scalars that did not fit the SGPR file (v_readlane / v_writelane) -- a static count, good for comparing two builds of a kernel.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only rodio_amd/csrc/rh_limit.hip -o /tmp/rh_limit.s
    python tools/isa_mix.py /tmp/rh_limit.s k_limit_scanILi2ELi16ELi8ELb0ELi0E
"""
import collections
import re
import sys


def main(path, pat):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(pat) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ops = collections.Counter()
    for l in lines[start:end]:
        m = re.match(r"\s+([a-z_0-9]+)(\s|$)", l)
        if m and not l.strip().startswith((".", ";")):
            ops[m.group(1)] += 1
    cls = collections.Counter()
    for k, v in ops.items():
        c = "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        cls[c] += v
    spill = ops["v_readlane_b32"] + ops["v_writelane_b32"]
    info = {}
    for l in lines[end:end + 60]:
        m = re.match(r"\s*\.set \S+\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
        if m:
            info[m.group(1)] = int(m.group(2))
    print(f"{pat}: {sum(ops.values())} instructions; valu {cls['valu']} (readlane/writelane {spill}, v_mov {ops['v_mov_b32_e32'] + ops['v_mov_b32_dpp']}, cndmask {ops['v_cndmask_b32_e64'] + ops['v_cndmask_b32_e32']}, "
          f"max {ops['v_max_f32_e32']}, pk {ops['v_pk_mul_f32'] + ops['v_pk_fma_f32'] + ops['v_pk_add_f32']}, trans {ops['v_log_f32_e32'] + ops['v_exp_f32_e32']}) salu {cls['salu']} (s_nop {ops['s_nop']}) lds {cls['lds']} vmem {cls['vmem']} | {info}")


if __name__ == "__main__":
    for pat in sys.argv[2:]:
        main(sys.argv[1], pat)
