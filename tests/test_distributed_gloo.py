"""World-size-2 gloo processes on CPU: the shard arithmetic, the equal and the ragged gather layout and the sharded Merkle reduction of
the multi-GPU path, restated on torch.distributed in tests/gloo_model.py (test support).  The per-rank engine here is the oracle's C
restatement, so that no GPU is needed.  It does NOT test the product's multi-rank path itself: that is pmx_mgpu_* in the C ABI, whose
every world > 1 branch runs on a GPU behind the stand-in collective library (tests/test_gpu_mgpu_standin.py: in-process slots, one
process per rank, and bench.py's own `--gpus N` forms) and on real RCCL with as many ranks as the box has GPUs (tests/test_gpu_mgpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import sponge_amd as S
import gloo_model as D
from sponge_amd import synth
from oracle import cref

from helpers import oracle_config

WORLD = 2
N_PER_RANK = 256
SEED = 0x5EED0004


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        f = S.BLS12_381_FR
        cr = cref.CRef(oracle_config("bls_t3_a5_8_31"))
        t = 3
        start, count = D.shard_bounds(WORLD * N_PER_RANK, WORLD, rank)
        assert count == N_PER_RANK and start == rank * N_PER_RANK
        # 1. sharded permutation + final gather
        shard = synth.random_elements(f, count * t, SEED, offset=start * t).reshape(count, t, 4)
        out_local = cr.permute_batch(shard, threads=1)
        gathered = D.all_gather_equal(torch.from_numpy(out_local.view(np.int64)))
        # 2. sharded Merkle root
        leaves = synth.random_elements(f, count, SEED + 1, offset=start)

        def subtree_root(x: torch.Tensor) -> torch.Tensor:
            nodes = cr.merkle(x.numpy().view(np.uint64).reshape(-1, 4), threads=1)
            return torch.from_numpy(nodes[-1].view(np.int64).copy())

        root = D.merkle_root_sharded(torch.from_numpy(leaves.view(np.int64)), subtree_root)
        # 3. ragged shards (2 * N_PER_RANK + 1 states over 2 ranks): bench.py's fallback gather, every rank's copy checked
        n_ragged = WORLD * N_PER_RANK + 1
        r_start, r_count = D.shard_bounds(n_ragged, WORLD, rank)
        mine = synth.random_elements(f, r_count * t, SEED + 2, offset=r_start * t).reshape(r_count, t, 4)
        whole = torch.zeros((n_ragged, t, 4), dtype=torch.int64)
        D.all_gather_rows(torch.from_numpy(cr.permute_batch(mine, threads=1).view(np.int64)), whole, host_staged=True)
        results[f"ragged{rank}"] = whole.numpy().view(np.uint64).copy()
        # 4. the last step and its gather piece by piece (pmx_mgpu_permute_gather_dev's bookkeeping): ragged shards, to every rank and to rank 1
        for to, chunks in ((-1, 3), (1, 4)):
            mine = torch.from_numpy(synth.random_elements(f, r_count * t, SEED + 2, offset=r_start * t).reshape(r_count, t, 4).view(np.int64).copy())
            receives = to < 0 or to == rank
            whole = torch.zeros((n_ragged, t, 4), dtype=torch.int64) if receives else None

            def step(piece: torch.Tensor) -> None:
                piece.copy_(torch.from_numpy(cr.permute_batch(piece.numpy().view(np.uint64), threads=1).view(np.int64)))

            D.gather_in_pieces(mine, whole, to, chunks, step)
            if receives:
                results[f"pieces{to}_{rank}"] = whole.numpy().view(np.uint64).copy()
        if rank == 0:
            results["gathered"] = gathered.numpy().view(np.uint64).copy()
            results["root"] = root.numpy().view(np.uint64).copy()
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_gather_and_merkle():
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), results), nprocs=WORLD, join=True)
    f = S.BLS12_381_FR
    cr = cref.CRef(oracle_config("bls_t3_a5_8_31"))
    whole = synth.random_elements(f, WORLD * N_PER_RANK * 3, SEED).reshape(WORLD * N_PER_RANK, 3, 4)
    assert np.array_equal(results["gathered"].reshape(-1, 3, 4), cr.permute_batch(whole, threads=2))
    leaves = synth.random_elements(f, WORLD * N_PER_RANK, SEED + 1)
    assert np.array_equal(results["root"].reshape(4), cr.merkle(leaves, threads=2)[-1])
    n_ragged = WORLD * N_PER_RANK + 1
    want = cr.permute_batch(synth.random_elements(f, n_ragged * 3, SEED + 2).reshape(n_ragged, 3, 4), threads=2)
    for rank in range(WORLD):
        assert np.array_equal(results[f"ragged{rank}"].reshape(-1, 3, 4), want), rank
        assert np.array_equal(results[f"pieces-1_{rank}"].reshape(-1, 3, 4), want), rank       # to every rank, three pieces
    assert np.array_equal(results["pieces1_1"].reshape(-1, 3, 4), want) and "pieces1_0" not in results   # to rank 1 only, four pieces


@pytest.mark.parametrize("count,chunks", [(0, 4), (1, 16), (7, 3), (16, 16), (100003, 8), (65536, 5)])
def test_pieces_partition_a_shard(count, chunks):
    spans = [D.piece_span(count, chunks, i) for i in range(chunks)]
    assert spans[0][0] == 0 and sum(c for _, c in spans) == count
    for (s0, c0), (s1, _) in zip(spans, spans[1:]):
        assert s0 + c0 == s1
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


@pytest.mark.parametrize("n,world", [(10, 3), (8, 8), (7, 8), (1 << 24, 8), (5, 1)])
def test_shard_bounds_partition(n, world):
    spans = [D.shard_bounds(n, world, r) for r in range(world)]
    assert spans[0][0] == 0 and sum(c for _, c in spans) == n
    for (s0, c0), (s1, _) in zip(spans, spans[1:]):
        assert s0 + c0 == s1
    assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
