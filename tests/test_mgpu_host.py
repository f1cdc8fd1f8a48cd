"""CPU-side checks of the device-group entry points (pmx_mgpu_*): shard arithmetic - the part of the multi-GPU path
that is pure host code - and the loud failure without a device.  No kernel and no RCCL call runs here."""
import ctypes

import numpy as np
import pytest

import sponge_amd as S
from sponge_amd import _lib, mgpu
import gloo_model as D
from sponge_amd.poseidon import c_config


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, (1 << 24), (1 << 24) + 5, (1 << 40) + 3])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8, 16])
def test_shard_bounds_partition(n, world):
    spans = [mgpu.shard_bounds(n, world, r) for r in range(world)]
    assert spans[0][0] == 0 and sum(c for _, c in spans) == n
    for (s0, c0), (s1, _) in zip(spans, spans[1:]):
        assert s0 + c0 == s1                                   # contiguous, in rank order
    counts = [c for _, c in spans]
    assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)   # the first n % world one longer
    # the torch.distributed helper used by the gloo tests partitions identically
    assert spans == [D.shard_bounds(n, world, r) for r in range(world)]


def test_c4_and_c5_shards_are_the_baseline_sizes():
    # BASELINE configs[3]: 2^24 states on 8 GPUs -> 2^21 per GPU; configs[4]: 2^24 leaves -> 2^21-leaf subtrees
    assert [mgpu.shard_bounds(1 << 24, 8, r) for r in range(8)] == [(r << 21, 1 << 21) for r in range(8)]


@pytest.mark.parametrize("world,rank", [(0, 0), (-1, 0), (4, 4), (4, -1)])
def test_shard_bounds_rejects_bad_ranks(world, rank):
    with pytest.raises(S.PmxError) as ei:
        mgpu.shard_bounds(10, world, rank)
    assert ei.value.code == _lib.PMX_ERR_ARG


def test_group_creation_without_a_device_fails_loudly():
    if _lib.lib().pmx_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = S.poseidon_config_from_lfsr(S.BLS12_381_FR, 2, 5, 8, 31)
    with pytest.raises(S.PmxError) as ei:
        mgpu.DeviceGroup.single_process(cfg, 1)
    assert ei.value.code == _lib.PMX_ERR_HIP and "no CPU fallback" in str(ei.value)
    with pytest.raises(S.PmxError) as ei:
        mgpu.DeviceGroup.one_rank(cfg, 0, 0, 1, bytes(_lib.UNIQUE_ID_BYTES))
    assert ei.value.code == _lib.PMX_ERR_HIP
    # the shared-context entry fails the same way (no silent CPU context)
    h = ctypes.c_void_p()
    c = c_config(cfg)
    assert _lib.lib().pmx_ctx_acquire(ctypes.byref(c), 0, ctypes.byref(h)) == _lib.PMX_ERR_HIP and not h.value


def test_null_arguments():
    lib = _lib.lib()
    assert lib.pmx_mgpu_create(None, 1, None, None) == _lib.PMX_ERR_ARG
    assert lib.pmx_mgpu_get_info(None, None) == _lib.PMX_ERR_ARG
    assert lib.pmx_mgpu_synchronize(None) == _lib.PMX_ERR_ARG
    assert lib.pmx_mgpu_destroy(None) == _lib.PMX_OK
    assert lib.pmx_ctx_release(None) == _lib.PMX_OK
    assert lib.pmx_ctx_cache_clear() == _lib.PMX_OK
    assert not lib.pmx_mgpu_stream(None, 0) and not lib.pmx_mgpu_ctx(None, 0)


def test_bench_gpus_n_without_a_launcher_starts_its_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must not answer with a usage message (rounds 1-4 did): it starts
    its ranks itself, as a CHILD process, and exits with the child's code.  Without a GPU every rank refuses ("no CPU fallback") and the
    code is not zero - what this CPU test pins is the launcher: the child command, no hang, no exec, the failure carried back.  The same
    command on one GPU behind the stand-in collective library, with its JSON line checked: tests/test_gpu_mgpu_standin.py."""
    import os
    import subprocess
    import sys
    if _lib.lib().pmx_device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_mgpu_standin.py runs the real thing")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = p.stderr.decode(errors="replace")
    assert p.returncode != 0 and not p.stdout.strip(), (p.returncode, p.stdout[-500:])
    assert "launching the ranks as a child process" in err and "torch.distributed.run" in err and "--nproc-per-node=2" in err
    assert "needs an MI355X" in err, err[-2000:]
    # ... and the one-process form says the same without starting anything
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-process"], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode != 0 and b"needs an MI355X" in p.stderr
