"""Source sharding of the mixer across the GPUs of one node.

rodio's streams only meet in the mixer sum (/root/reference/src/mixer.rs:185-198); every stage
before it touches one source.  So rank r of R owns a contiguous shard of the sources, runs the
fused kernel on it for the same output time range, and the partial mixes are summed with ONE
all-reduce per mixed block (RCCL over xGMI when the process group is "nccl"; gloo on CPU in the
tests).  No other communication.  One process per GPU, torch.distributed for the collective.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_range(n_sources: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the sources rank `rank` owns: contiguous, sizes differ by at most one, the first
    n_sources % world ranks take the extra source (insertion order is kept inside a shard)."""
    if world <= 0 or not 0 <= rank < world or n_sources < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_sources, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(items: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_range(len(items), rank, world)
    return list(items[lo:hi])


def all_reduce_mix(block, group=None, async_op: bool = False):
    """Sum the ranks' partial mixes of one block in place (every rank ends up with the full mix, like
    rodio's single MixerSource).  `block` is a contiguous f32 tensor holding the partial mix of this
    rank for the SAME output frames on every rank.  Returns the work handle when async_op."""
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(block, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def max_out_frames(out_frames: int, group=None) -> int:
    """Shards may hold sources of different lengths: the mixed block is as long as the longest source
    of ANY rank, so the ranks agree on the length (and zero-pad their partial mix) before reducing."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return out_frames
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([out_frames], dtype=torch.int64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


class ShardedResampleLowpassMix:
    """`ResampleLowpassMix` over this rank's shard of the sources + the cross-rank mixer sum.

    set_sources() takes ALL sources of the job (device tensors of this rank for its own shard are
    enough: entries outside the shard may be None) and keeps shard_range(len, rank, world)."""

    def __init__(self, rank: int, world: int, *args, group=None, **kw):
        from .source import ResampleLowpassMix

        self.rank, self.world, self.group = rank, world, group
        self.check = kw.pop("check_status", True)  # synchronise and read the shard's status word in front of the collective
        self.pipe = ResampleLowpassMix(*args, **kw)
        self.out_frames = 0

    def set_sources(self, tensors):
        mine = shard(tensors, self.rank, self.world)
        self.pipe.set_sources(mine)
        self.out_frames = max_out_frames(self.pipe.out_frames, self.group)

    def run(self, out=None, async_op: bool = False):
        """One block: local fused kernel, then the all-reduce.  Returns (mixed, work)."""
        import torch

        ch = self.pipe.channels
        if out is None:
            out = torch.empty(max(self.out_frames * ch, 4), device="cuda", dtype=torch.float32)
        local = self.pipe.run(out)
        if self.check:  # a hand-off that timed out poisons its tile with NaN: it must not reach the other ranks' mixes as well
            self.pipe.check_status()
        if local.numel() < self.out_frames * ch:  # this shard's sources are shorter than another rank's
            out[local.numel(): self.out_frames * ch].zero_()
        mixed = out[: self.out_frames * ch]
        return mixed, all_reduce_mix(mixed, self.group, async_op)


class NativeComm:
    """The C-ABI communicator (rh_comm_*: RCCL without PyTorch in the loop) -- what a Rust host uses.
    Rank 0 creates the id with NativeComm.unique_id() and ships the 128 bytes to the other ranks."""

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C

        from ._lib import check, lib

        buf = (C.c_uint8 * 128)()
        check(lib.rh_comm_unique_id(buf), "rh_comm_unique_id")
        return bytes(buf)

    def __init__(self, rank: int, nranks: int, uid: bytes):
        import ctypes as C

        from ._lib import check, lib

        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        check(lib.rh_comm_init(C.byref(self._h), rank, nranks, buf), "rh_comm_init")

    def all_reduce(self, block, stream=None):
        import ctypes as C

        import torch

        from ._lib import check, lib

        s = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        check(lib.rh_allreduce_sum_f32(self._h, C.c_void_p(block.data_ptr()), block.numel(), s), "rh_allreduce_sum_f32")

    def reduce(self, block, root=0, stream=None):
        import ctypes as C

        import torch

        from ._lib import check, lib

        s = C.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)
        check(lib.rh_reduce_sum_f32(self._h, C.c_void_p(block.data_ptr()), block.numel(), root, s), "rh_reduce_sum_f32")

    def close(self):
        from ._lib import lib

        if self._h:
            lib.rh_comm_destroy(self._h)
            self._h = None
