"""Multi-GPU plumbing: one process per GPU, pose batches sharded contiguously, ONE gather of the
solved transforms.  The reference has no multi-device code at all (test.cpp:14 has
cudaSetDevice commented out); hypotheses are independent, so the only exchange is the final
gather of P x 72-byte RegistrationResult records -- RCCL over xGMI on the GPUs
(torch.distributed backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

RESULT_FLOATS = 18          # RegistrationResult = 16 f32 transform + rmse + fitness (icp.h:33-35)


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of rank `rank`: (first, count); the first n_items % world ranks get one extra.
    Same rule as pr_shard_range in the C ABI."""
    world = max(1, world)
    if rank >= world:
        return n_items, 0                                          # no such rank: an empty block behind the last one
    base, extra = divmod(n_items, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


class GatherWork:
    """Handle of an in-flight gather: .wait() completes it, .out is the list of per-rank tensors on dst (None elsewhere)."""

    def __init__(self, work, out, keep=None):
        self._work, self.out, self._keep = work, out, keep     # keep: the (possibly padded) send buffer must outlive the op

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return True


def gather_results(local, world: int, rank: int, dst: int = 0, max_count: Optional[int] = None, async_op: bool = False):
    """Gather per-rank result tensors (count_r x 18 float32, on the device the process group uses)
    to `dst` with a single collective.  Shards may differ by one record, so every rank pads to
    `max_count` records; returns the list of per-rank padded tensors on dst, None elsewhere.
    With async_op=True returns a work handle with .wait() (and .out = the list on dst)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return GatherWork(None, [local]) if async_op else [local]
    n_local = local.numel() // RESULT_FLOATS
    if max_count is None:
        # shards of a batch differ by at most one record (shard_bounds): every rank must send the same number of elements, so
        # agree on the largest shard first (one 8-byte all-reduce; pass max_count = ceil(n_items / world) to skip it)
        cap_t = torch.tensor([n_local], dtype=torch.int64, device=local.device)
        dist.all_reduce(cap_t, op=dist.ReduceOp.MAX)
        cap = int(cap_t.item())
    else:
        cap = max_count
        if cap < n_local:
            raise ValueError(f"max_count {cap} is smaller than this rank's {n_local} records")
    buf = local.reshape(-1)
    if cap != n_local:
        buf = torch.zeros(cap * RESULT_FLOATS, dtype=local.dtype, device=local.device)
        buf[: n_local * RESULT_FLOATS] = local.reshape(-1)
    out: Optional[List] = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    if async_op:
        return GatherWork(dist.gather(buf, out, dst=dst, async_op=True), out, keep=buf)
    dist.gather(buf, out, dst=dst)
    return out


def assemble(gathered, n_items: int, world: int):
    """Concatenate the gathered (padded) shards back into global hypothesis order."""
    import torch
    parts = []
    for r in range(world):
        _, cnt = shard_bounds(n_items, r, world)
        parts.append(gathered[r].reshape(-1, RESULT_FLOATS)[:cnt])
    return torch.cat(parts, dim=0)
