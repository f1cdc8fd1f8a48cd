"""TEST SUPPORT (round 6: moved out of the package - nothing in the product or in bench.py uses it): a torch.distributed model of the
sharded batch for the world-size-2 gloo tests on CPU (tests/test_distributed_gloo.py).  Sponge states are independent (reference
src/poseidon/mod.rs:62-183 has no cross-state data flow), so the batch is cut into contiguous shards and there is NO collective on the
data path; a collective is used only for the final gather of results (and for the 32-byte subtree roots of the Merkle mode).

The product's multi-GPU path is the C ABI's device group (pmx_mgpu_*, sponge_amd/mgpu.py), which talks to RCCL itself and partitions
with the same arithmetic (pmx_shard_bounds); these helpers restate that partition and the two gathers on torch.distributed so that the
shard arithmetic and the gather layout can be exercised by real ranks where there is no GPU.
"""
from __future__ import annotations

from typing import Callable, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous partition: rank g owns [start, start+count); the first n_total % world ranks get one more."""
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def all_gather_equal(local: torch.Tensor, out: torch.Tensor | None = None, async_op: bool = False):
    """All-gather of equally sized shards into [world * n, ...] in rank order (ring/direct all-gather by RCCL)."""
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend() == "gloo":
        # rehearsal of the multi-GPU path on a box without RCCL peers: stage through the host (synchronous)
        parts = [torch.empty_like(local, device="cpu") for _ in range(world)]
        dist.all_gather(parts, local.cpu())
        out.view(-1).copy_(torch.cat([p.reshape(-1) for p in parts]))
        work = _Done()
    else:
        work = dist.all_gather_into_tensor(out.view(-1), local.contiguous().view(-1), async_op=async_op)
    return (out, work) if async_op else out


def all_gather_rows(local: torch.Tensor, out: torch.Tensor, host_staged: bool = False) -> None:
    """Rank r's rows land at r's span of `out` ([n_total, ...]; spans = shard_bounds(n_total, world, r)).  Shards may be
    ragged: every rank pads to the longest one.  host_staged: gather CPU copies (gloo cannot move device tensors)."""
    world = dist.get_world_size()
    spans = [shard_bounds(out.shape[0], world, r) for r in range(world)]
    longest = max(c for _, c in spans)
    row = tuple(local.shape[1:])
    assert local.shape[0] == spans[dist.get_rank()][1], "this rank's shard has the wrong length"
    padded = local
    if local.shape[0] != longest:
        padded = torch.zeros((longest,) + row, dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    if host_staged:
        parts = [torch.empty((longest,) + row, dtype=local.dtype) for _ in range(world)]
        dist.all_gather(parts, padded.cpu())
    else:
        flat = torch.empty((world * longest,) + row, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(flat.view(-1), padded.contiguous().view(-1))
        parts = [flat[r * longest:(r + 1) * longest] for r in range(world)]
    for r, (s_r, c_r) in enumerate(spans):
        out[s_r:s_r + c_r].copy_(parts[r][:c_r])


class _Done:
    def wait(self):
        return True


def merkle_root_sharded(leaves_local: torch.Tensor, subtree_root: Callable[[torch.Tensor], torch.Tensor],
                        ) -> torch.Tensor:
    """2-to-1 Merkle root of world * m leaves, m a power of two per rank, world a power of two.
    Every rank reduces its own contiguous subtree with `subtree_root` (the GPU tree kernel), the 32-byte
    subtree roots are all-gathered, and every rank finishes the top log2(world) levels with the same function."""
    world = dist.get_world_size()
    assert world & (world - 1) == 0, "world size must be a power of two"
    root_local = subtree_root(leaves_local).reshape(1, 4)
    if world == 1:
        return root_local.reshape(4)
    roots = all_gather_equal(root_local)              # [world][4], rank order = leaf order
    return subtree_root(roots).reshape(4)


def piece_span(count: int, chunks: int, i: int) -> Tuple[int, int]:
    """Piece i of `chunks` of a shard of `count` units (pmx_mgpu.cpp: piece_span): [first, first + cnt), pieces differ by at most one unit."""
    lo = count // chunks * i + count % chunks * i // chunks
    hi = count // chunks * (i + 1) + count % chunks * (i + 1) // chunks
    return lo, hi - lo


def gather_in_pieces(local: torch.Tensor, out: torch.Tensor | None, root: int, chunks: int, step: Callable[[torch.Tensor], None]) -> None:
    """pmx_mgpu_permute_gather_dev restated: the last step on `local` ([count, ...], in place through `step`) in `chunks` pieces, piece i's
    transfers - to `root`, or to every rank when root < 0 - posted right behind piece i's step as point-to-point messages; `out`
    ([n_total, ...]) is written on the receivers only.  Every rank derives every peer's piece from (n_total, world, chunks, i) alone."""
    world, me = dist.get_world_size(), dist.get_rank()
    n_total = out.shape[0] if out is not None else None
    if n_total is None:                       # a rank that does not receive still needs the total to know nothing about the others: it only sends
        assert root >= 0 and root != me
    i_receive = root < 0 or root == me
    pending = []
    for i in range(chunks):
        pf, pc = piece_span(local.shape[0], chunks, i)
        if pc:
            step(local[pf:pf + pc])
        ops = []
        if i_receive:
            start, _ = shard_bounds(n_total, world, me)
            out[start + pf:start + pf + pc].copy_(local[pf:pf + pc])
        for peer in range(world):
            if peer == me:
                continue
            if (root < 0 or root == peer) and pc:
                ops.append(dist.P2POp(dist.isend, local[pf:pf + pc].contiguous(), peer))
            if i_receive:
                qs, qn = shard_bounds(n_total, world, peer)
                qf, qc = piece_span(qn, chunks, i)
                if qc:
                    ops.append(dist.P2POp(dist.irecv, out[qs + qf:qs + qf + qc], peer))
        if ops:
            pending += dist.batch_isend_irecv(ops)      # (the next piece's step runs while these are in flight)
    for w in pending:
        w.wait()
