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
LD_LIBRARY_PATH, and the shared-device group needs a hook that only libposeidon_mi355x_test.so has (the shipped objects with
pmx_mgpu.cpp compiled -DPMX_TEST_HOOKS; the workers bind that build, the library that ships has no such code).  The reference has no counterpart (src/poseidon/mod.rs:62-183: states are independent); expected
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
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    return env


@pytest.mark.parametrize("world", [2, 3, 8, 16])
def test_every_multi_rank_branch_on_one_gpu(world, tmp_path):
    _ensure_fake()
    out = str(tmp_path / f"standin_w{world}.json")
    p = subprocess.run([sys.executable, os.path.join(HERE, "mgpu_standin_worker.py"), str(world), out], env=_child_env(FAKE_DIR),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert os.path.exists(out), p.stdout.decode(errors="replace")[-3000:]
    res = json.load(open(out))
    assert res["error"] is None, res["error"]
    failed = [s for s in res["scenarios"] if not s["ok"]]
    assert not failed, "\n\n".join(f"{s['name']}:\n{s['detail']}" for s in failed)
    names = {s["name"] for s in res["scenarios"]}
    assert {"host_batch_fan_out", "fan_out_failure_on_a_nonzero_slot_and_serial_fallback", "gather_equal_shards_is_one_all_gather",
            "gather_ragged_shards_is_one_broadcast_per_rank", "gather_with_empty_shards", "gather_to_one_rank_is_world_minus_one_messages",
            "gather_to_one_rank_with_empty_shards_and_bad_arguments", "last_step_with_its_gather_piece_by_piece",
            "stand_in_refuses_a_send_nobody_receives", "wide_states_gathered_piece_by_piece_and_to_one_rank", "sharded_merkle_host_leaves",
            "sharded_merkle_device_resident_every_rank_holds_the_whole_top", "one_group_per_rank_create_rank",
            "collective_failures_are_status_codes", "every_communicator_was_destroyed"} <= names
    assert res["ok"] and p.returncode == 0


def test_a_collective_library_without_a_needed_symbol_is_reported_not_crashed(tmp_path):
    """pmx_mgpu.cpp binds fourteen entry points by name; a librccl that lacks one (here: ncclBroadcast) must turn into
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


def _xproc_env():
    """ranks in different processes: the test-hook build of the library is told WHICH collective library to bind
    (PMX_RCCL_LIBRARY; the shipped library does not read it) and the stand-in is told that its ranks are processes"""
    return dict(os.environ, PMX_RCCL_LIBRARY=os.path.join(FAKE_DIR, "librccl.so.1"), FAKE_RCCL_XPROC="1", HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.mark.parametrize("world,shape", [(2, "equal"), (2, "ragged"), (3, "ragged"), (4, "equal"), (8, "ragged")])
def test_one_process_per_rank_on_one_gpu(world, shape, tmp_path):
    """The multi-PROCESS form - what `bench.py --gpus N` and a Rust job with one process per GPU use, and what
    tests/test_gpu_mgpu.py::test_one_process_per_gpu_with_create_rank runs with ONE rank on a one-GPU box: W fresh child
    processes, all on cuda:0, the id made by rank 0 (pmx_mgpu_unique_id) and handed over through a file,
    pmx_mgpu_create_rank in every child, sharded permutation, equal (one ncclAllGather) or ragged (W grouped in-place
    ncclBroadcasts, the last shards shorter) gather, the sharded tree with its roots all-gather; every rank checks the WHOLE
    of its gathered copy and its tree top against the C restatement (tests/mgpu_rank_worker.py, unchanged)."""
    _ensure_fake()
    n_total = (1 << 12) * world + (3 if shape == "ragged" else 0)
    uid_file = str(tmp_path / "uid.bin")
    procs = []
    for r in range(world):
        out = str(tmp_path / f"rank{r}.json")
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(HERE, "mgpu_rank_worker.py"), str(r), str(world), "0", uid_file,
                                             str(n_total), "8", out], env=_xproc_env(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    results = []
    for out, p in procs:
        try:
            log, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for _, q in procs:
                q.kill()
            raise
        assert os.path.exists(out), log.decode(errors="replace")[-3000:]
        results.append(json.load(open(out)))
    for res in results:
        assert res["ok"], res
        assert res["info"]["comm_ranks"] == world and res["gather_bad_spans"] == [] and res["p2p_bad"] == []
        if world & (world - 1) == 0:
            assert res["tree_ok"] is True
    assert sorted(r["info"]["comm_first_rank"] for r in results) == list(range(world))
    leftovers = [f for f in os.listdir("/dev/shm") if f.startswith("fake_rccl_")]
    assert not leftovers, leftovers


def test_a_named_collective_library_that_does_not_exist_is_reported(tmp_path):
    """(test-hook build) PMX_RCCL_LIBRARY names the collective library to bind; a path that cannot be loaded is PMX_ERR_RCCL with the
    loader's reason at the first device-group call (nothing else is tried: a caller who names a library wants THAT library)."""
    code = r'''
import sys
sys.path.insert(0, %r)
import sponge_amd as S
from sponge_amd import _lib, mgpu
_lib.use_test_library()          # (the shipped library does not read PMX_RCCL_LIBRARY)
try:
    mgpu.unique_id()
    sys.exit(2)
except S.PmxError as e:
    assert e.code == _lib.PMX_ERR_RCCL and "/nonexistent/librccl.so could not be loaded" in str(e), str(e)
''' % os.path.dirname(HERE)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PMX_RCCL_LIBRARY="/nonexistent/librccl.so"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-3000:]


@pytest.mark.parametrize("world,extra", [(2, ["--workload", "c2", "--total-units", "50001", "--gather", "step"]),
                                         (3, ["--workload", "c2", "--total-units", "50001", "--gather", "root"]),
                                         (3, ["--workload", "c2", "--total-units", "50001", "--gather", "overlap", "--gather-chunks", "5"]),
                                         (2, ["--workload", "c2", "--total-log2", "15", "--gather", "overlap-root"]),
                                         (4, ["--workload", "c5", "--total-log2", "14"])])
def test_bench_multi_rank_path_runs_through_the_device_group(world, extra, tmp_path):
    """`bench.py --gpus N` under torch.distributed.run, N > 1, on one GPU (PMX_BENCH_REHEARSAL=group): the ranks share
    cuda:0, torch.distributed's control plane is gloo, and the data path is the product's - pmx_mgpu_create_rank with the id
    broadcast by rank 0, pmx_mgpu_permute_shards_dev, the (ragged) pmx_mgpu_all_gather_dev per step, the sharded tree with its
    roots all-gather - followed by bench.py's own verification of every rank's gathered copy.  The JSON line must say
    rccl.ranks = N and verified = true; the numbers are a rehearsal's (tools/gpu_group_rehearsal.sh runs more shapes)."""
    _ensure_fake()
    root = os.path.dirname(HERE)
    env = dict(_xproc_env(), PMX_BENCH_REHEARSAL="group", MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() * 7 + world) % 400
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--spinup-seconds", "0.05"] + extra, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    text = p.stdout.decode(errors="replace")
    lines = [json.loads(l) for l in text.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, text[-3000:]
    d = lines[0]
    assert d["verified"] is True and d["n_gpus"] == world and d["rccl"]["ranks"] == world, d
    assert "pmx_mgpu_create_rank" in d["rccl"]["via"] and "REHEARSAL" in d["config"]


@pytest.mark.parametrize("world,form,extra", [(2, "ranks", ["--workload", "c2", "--total-log2", "16"]),
                                              (8, "ranks", ["--workload", "c2", "--total-units", "100003"]),
                                              (2, "single", ["--workload", "c2", "--total-units", "50001"]),
                                              (8, "single", ["--workload", "c2", "--total-log2", "17"]),
                                              (3, "single", ["--workload", "c2", "--total-units", "50001", "--gather", "root"]),
                                              (4, "single", ["--workload", "c2", "--total-units", "50001", "--gather", "overlap"]),
                                              (3, "single", ["--workload", "c2", "--total-log2", "15", "--gather", "overlap-root", "--gather-chunks", "3"]),
                                              (8, "single", ["--workload", "c5", "--total-log2", "17"]),
                                              (4, "ranks", ["--workload", "c5", "--total-log2", "14"])])
def test_bench_gpus_n_needs_no_launcher(world, form, extra, tmp_path):
    """`python bench.py --gpus N` exactly as the driver spells it at N = 1 - no torch.distributed.run around it, no WORLD_SIZE in the
    environment - must produce its ONE JSON line by itself.  form "ranks": bench.py starts `python -m torch.distributed.run ...` as a
    CHILD process (before torch is imported, never exec) and exits with its code; form "single": --single-process, one process drives
    all N device slots through pmx_mgpu_create (ncclCommInitAll).  On this one-GPU box the N ranks / slots share cuda:0 behind the
    stand-in collective library (PMX_BENCH_REHEARSAL=group: test-hook build of the library): rc 0, one line, rccl.ranks = N,
    verified = true (every rank's / device's gathered copy or tree top against the C restatement)."""
    _ensure_fake()
    root = os.path.dirname(HERE)
    env = dict(_xproc_env(), PMX_BENCH_REHEARSAL="group")
    if form == "single":
        env.pop("FAKE_RCCL_XPROC")                     # the slots are threads of one process here
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1", "--spinup-seconds", "0.05"] + extra
    if form == "single":
        cmd.append("--single-process")
    p = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    text, err = p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    lines = [json.loads(l) for l in text.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (text[-2000:], err[-3000:])
    d = lines[0]
    assert d["verified"] is True and d["n_gpus"] == world and d["rccl"]["ranks"] == world, d
    assert d["cpu_baseline"] is None and "REHEARSAL" in d["config"]
    overlapped = any(x in ("overlap", "overlap-root") for x in extra)
    assert (d["gather_ms"] is None and d["gather_inside_last_step"] is True) if overlapped else d["gather_ms"] is not None
    if form == "single":
        assert "pmx_mgpu_create (ncclCommInitAll" in d["rccl"]["via"] and "launcher" in d["config"]
    else:
        assert "pmx_mgpu_create_rank" in d["rccl"]["via"] and "launching the ranks as a child process" in err
