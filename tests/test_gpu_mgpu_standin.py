"""World > 1 on ONE GPU: every multi-rank branch of the device-group code (sponge_amd/csrc/pmx_mgpu.cpp) through the real
C ABI, with W = 2, 3, 8, 16 "ranks" sharing cuda:0 behind a stand-in for librccl.so.1 (tests/fake_rccl: collectives performed
as device-to-device copies on the caller's streams, every range and every rank's call sequence checked).

Why: the boxes this is developed on have one GPU and RCCL refuses two ranks on one device, so BASELINE's configs[3]
(2^24 states over 8 GPUs + gather) and configs[4] (2^24-leaf tree over 8 GPUs) had never executed any index arithmetic
past world = 1.  What this proves: the product's own bookkeeping - shard offsets and counts, the in-place / `mine` logic
of the ragged gather, the d_top layout and top levels of the sharded tree, the host fan-out with W - 1 worker threads
and its error carry-back, one-group-per-rank with first_rank != 0.  What it cannot prove: RCCL itself and xGMI
(tests/test_gpu_mgpu.py runs the same entry points on real RCCL with as many ranks as the box has GPUs).

The stand-in is test infrastructure: it is found only because the child process below gets tests/fake_rccl first on
LD_LIBRARY_PATH; the product is unchanged (the shared-device group needs the PMX_TEST_HOOKS=1 hook, which is inert in
any other process).  The reference has no counterpart (src/poseidon/mod.rs:62-183: states are independent); expected
values are the C restatement's over the whole batch / tree."""
import json
import os
import subprocess
import sys

import pytest

from sponge_amd import _lib

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FAKE_DIR = os.path.join(HERE, "fake_rccl")


def _ensure_fake():
    if not (os.path.exists(os.path.join(FAKE_DIR, "librccl.so.1")) and os.path.exists(os.path.join(FAKE_DIR, "broken", "librccl.so.1"))):
        subprocess.check_call(["make", "-C", FAKE_DIR, "all"])


def _child_env(libdir):
    env = dict(os.environ, PMX_TEST_HOOKS="1")
    env["LD_LIBRARY_PATH"] = libdir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    return env


@pytest.mark.parametrize("world", [2, 3, 8, 16])
def test_every_multi_rank_branch_on_one_gpu(world, tmp_path):
    _ensure_fake()
    out = str(tmp_path / f"standin_w{world}.json")
    p = subprocess.run([sys.executable, os.path.join(HERE, "mgpu_standin_worker.py"), str(world), out], env=_child_env(FAKE_DIR),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert os.path.exists(out), p.stdout.decode(errors="replace")[-3000:]
    res = json.load(open(out))
    assert res["error"] is None, res["error"]
    failed = [s for s in res["scenarios"] if not s["ok"]]
    assert not failed, "\n\n".join(f"{s['name']}:\n{s['detail']}" for s in failed)
    names = {s["name"] for s in res["scenarios"]}
    assert {"host_batch_fan_out", "fan_out_failure_on_a_nonzero_slot_and_serial_fallback", "gather_equal_shards_is_one_all_gather",
            "gather_ragged_shards_is_one_broadcast_per_rank", "gather_with_empty_shards", "sharded_merkle_host_leaves",
            "sharded_merkle_device_resident_every_rank_holds_the_whole_top", "one_group_per_rank_create_rank",
            "collective_failures_are_status_codes", "every_communicator_was_destroyed"} <= names
    assert res["ok"] and p.returncode == 0


def test_a_collective_library_without_a_needed_symbol_is_reported_not_crashed(tmp_path):
    """pmx_mgpu.cpp binds twelve entry points by name; a librccl that lacks one (here: ncclBroadcast) must turn into
    PMX_ERR_RCCL naming the symbol at the first device-group call - and the single-device ABI keeps working."""
    _ensure_fake()
    code = r'''
import ctypes, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import sponge_amd as S
from sponge_amd import _lib, mgpu, synth
from gpu_helpers import c_oracle, product_config
cfg = product_config("bls_t3_a5_8_31")
try:
    mgpu.DeviceGroup.single_process(cfg, 1)
    sys.exit(2)
except S.PmxError as e:
    assert e.code == _lib.PMX_ERR_RCCL and "librccl has no symbol ncclBroadcast" in str(e), str(e)
try:
    mgpu.unique_id()
    sys.exit(3)
except S.PmxError as e:
    assert e.code == _lib.PMX_ERR_RCCL, str(e)
st = synth.random_elements(cfg.field, 300, seed=5).reshape(100, 3, 4)
assert np.array_equal(cfg.context(0).permute_batch(st), c_oracle("bls_t3_a5_8_31").permute_batch(st, threads=0))
''' % (os.path.dirname(HERE), HERE)
    p = subprocess.run([sys.executable, "-c", code], env=_child_env(os.path.join(FAKE_DIR, "broken")), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]
