#!/usr/bin/env python3
"""Benchmark of the batched Poseidon permutation on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (PoseidonSponge::permute on every state, reference
src/poseidon/mod.rs:95-118) over one device-resident batch of synthetic random states.

  N = 1   BASELINE.json configs[1] (C2): 2^20 independent states, BLS12-381 Fr, t=3, alpha=5, 8+31 rounds;
          step = one pmx_permute_batch_dev launch.
  N > 1   (launched by torch.distributed.run, one rank per GPU) BASELINE.json configs[3] (C4): 2^24 states in
          all, sharded contiguously - 2^24 / N per GPU, strong scaling, NO collective on the data path; after the
          K steps the result shards are all-gathered once over RCCL inside the timed region (the "final gather").
          Every rank drives its shard through the C ABI's device group (pmx_mgpu_create_rank: ncclCommInitRank,
          pmx_mgpu_permute_shards_dev, pmx_mgpu_all_gather_dev); torch.distributed only carries the communicator
          id to the ranks and does the barrier / max-over-ranks of the timing contract.
  --workload c5   BASELINE.json configs[4]: 2-to-1 Merkle tree of 2^24 leaves in all (2^24 / N per GPU; subtree per
          rank, all-gather of the N 32-byte subtree roots, top log2 N levels on every rank).
  --workload c3 / h3 / h9   wide states (BN254 Fr, t=9, 2^18 per GPU) and the absorb/squeeze hash driver.

Rank 0 prints ONE JSON line.  `value` = permutations per second over all ranks, inputs already in HBM.
`roofline` prices the dominant kernel against HBM (algorithmic 2*t*32 bytes per permutation); `int_valu` against
the v_mad_u64_u32 issue rate measured on this device in this run (pmx_diag_int_valu_peak), which is what binds it;
`engine` is what the library's launchers dispatch for this call (pmx_ctx_engine_info) and feeds the multiply count;
`valu_issue` prices the kernel's VALU instruction count against the issue slot measured in this run (pmx_diag_issue_slot);
`cpu_baseline` times the C restatement of the reference algorithm (oracle/, kind "port") on the host cores for a
bounded sample of the same workload (rank 0, N=1 only).  After the timed region one extra untimed pass over a fresh
copy of the seeded batch is compared with the C restatement on a sample ("verified"); a mismatch exits non-zero.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec

WORKLOADS = {
    # name: (field, rate, alpha, RF, RP, log2 units at N=1, log2 TOTAL units at N>1 (None: N=1 size per GPU), seed, description)
    "c2": ("bls12_381_fr", 2, 5, 8, 31, 20, 24, 0x5EED0002, "bls12_381_fr t=3 alpha=5 RF=8 RP=31 permutation batch"),
    "c3": ("bn254_fr", 8, 5, 8, 57, 18, None, 0x5EED0003, "bn254_fr t=9 alpha=5 RF=8 RP=57 permutation batch"),
    "c5": ("bls12_381_fr", 2, 5, 8, 31, 24, 24, 0x5EED0005, "bls12_381_fr t=3 alpha=5 2-to-1 Merkle tree"),
    # the absorb/squeeze batch driver (pmx_hash_batch_dev): per row new; absorb(L); squeeze_native(1)
    "h3": ("bls12_381_fr", 2, 5, 8, 31, 20, None, 0x5EED0006, "bls12_381_fr t=3 alpha=5 hash of 4 elements -> 1 (2 permutations/row)"),
    "h9": ("bn254_fr", 8, 5, 8, 57, 18, None, 0x5EED0007, "bn254_fr t=9 alpha=5 hash of 8 elements -> 1 (1 permutation/row)"),
    # the duplex driver on device-resident sponges with explicit mode words (pmx_sponge_absorb_batch_dev + pmx_sponge_squeeze_batch_dev):
    # every step absorbs L elements into every sponge and squeezes K out of it, the sponges carried from step to step
    "d3": ("bls12_381_fr", 2, 5, 8, 31, 20, None, 0x5EED0008, "bls12_381_fr t=3 alpha=5 duplex driver: absorb(4) + squeeze(3) per sponge and step (4 permutations)"),
    "d9": ("bn254_fr", 8, 5, 8, 57, 18, None, 0x5EED0009, "bn254_fr t=9 alpha=5 duplex driver: absorb(11) + squeeze(9) per sponge and step (4 permutations)"),
    # the reference's own rate-2 default and the config of its only permutation KAT (src/test.rs:15, src/poseidon/mod.rs:376-399): alpha = 17
    "k3": ("bls12_381_fr", 2, 17, 8, 31, 20, None, 0x5EED0011, "bls12_381_fr t=3 alpha=17 RF=8 RP=31 permutation batch (the reference's default rate-2 parameters)"),
    # the other widths of the reference's default table (src/test.rs:14-31), for tuning the wide-state engines
    "w4": ("bls12_381_fr", 3, 5, 8, 56, 19, None, 0x5EED0014, "bls12_381_fr t=4 alpha=5 RF=8 RP=56 permutation batch"),
    "w5": ("bls12_381_fr", 4, 5, 8, 56, 19, None, 0x5EED0015, "bls12_381_fr t=5 alpha=5 RF=8 RP=56 permutation batch"),
    "w6": ("bls12_381_fr", 5, 5, 8, 57, 18, None, 0x5EED0016, "bls12_381_fr t=6 alpha=5 RF=8 RP=57 permutation batch"),
    "w7": ("bls12_381_fr", 6, 5, 8, 57, 18, None, 0x5EED0017, "bls12_381_fr t=7 alpha=5 RF=8 RP=57 permutation batch"),
    "w8": ("bls12_381_fr", 7, 5, 8, 57, 18, None, 0x5EED0018, "bls12_381_fr t=8 alpha=5 RF=8 RP=57 permutation batch"),
}
HASH_SHAPES = {"h3": (4, 1, 2), "h9": (8, 1, 1)}     # workload -> (in_len, out_len, permutations per row)
# workload -> (absorb length, squeeze length, permutations per sponge and step once the sponges are in their steady cycle:
# a sponge left Squeezing{1} by squeeze(rate + 1) permutes at once when it absorbs, again when its rate fills, and twice in the squeeze)
DUPLEX_SHAPES = {"d3": (4, 3, 4), "d9": (11, 9, 4)}
BASELINE_CONFIG = {("c2", False): "BASELINE.json configs[1] (C2)", ("c2", True): "BASELINE.json configs[3] (C4)",
                   ("c3", False): "BASELINE.json configs[2] (C3)", ("c5", False): "BASELINE.json configs[4] (C5) on one GPU",
                   ("c5", True): "BASELINE.json configs[4] (C5)"}


def mads_per_permutation(t, alpha, rf, rp, optimised, row_tables=False, lane_tables=False, mfma_dense=False, window=0):
    """v_mad_u64_u32 count of one permutation as implemented (pmx_field.hpp): product 81, square 45, reduction 81 limb
    products; one reduction per S-box step and per matrix row.  With shifted tables (tab_dot) a row of N constants
    costs 81 N + 18 instead of 81 N + 81 (`row_tables`) and an identity-lane update 81 + 18 + 9 instead of 162 + 9
    (`lane_tables`).  The optimised schedule carries the state scaled lane by lane so that one entry per row of every
    matrix but the last round's is exactly one (pmx_prepare.hpp: derive_opt_tables).  `mfma_dense`: the rows of the DENSE
    layers come from the matrix cores (pmx_mfma.hpp) - 8 multiplies each, the Montgomery step inside the word sums of the row's finish
    (the 40 v_mad_i64_i32 of the word sums themselves are not counted: the figure is about v_mad_u64_u32).
    `window` = K > 0: the partial rounds as windows of K S-boxes (the first window takes the remainder), each closed by one
    matrix-core layer of t rows; on the VALU a window keeps its S-boxes and the history products of its later S-box inputs
    (x_{k+1} = z_k + u_k + sum_{i<k} h_{k,i} z_i: a (k-1)-term dot product with an addend, element form); the layer after the
    entrance round is a matrix-core layer as well."""
    sqr, mul = 45 + 81, 81 + 81
    chain = {5: 2 * sqr + mul, 17: 4 * sqr + mul}.get(alpha)
    if chain is None:
        bits = bin(alpha)[3:]
        chain = len(bits) * sqr + bits.count("1") * mul
    red = 18 if row_tables == 1 or row_tables is True else 81
    dot = 81 * t + red                                      # a t-term row
    norm = 81 * (t - 1) + red + 9                           # a normalised row: t - 1 terms + 9 multiply-by-one injections of the addend
    lane = (81 + 18 + 9) if lane_tables else (mul + 9)
    sparse = norm + (t - 1) * lane                          # a sparse layer: row 0 and the identity lanes
    row_finish = 8
    if mfma_dense:
        dot = norm_dense = row_finish
    else:
        norm_dense = norm
    if optimised and mfma_dense and window > 0:
        n_win = -(-rp // window)
        sizes = [rp - (n_win - 1) * window] + [window] * (n_win - 1)
        if row_tables == 2:      # the history terms as rows on the matrix cores too (pmx_mfma.hpp: mfma_hist_rows): a row finish each
            hist = sum(row_finish for kw in sizes for k in range(2, kw))
        else:
            hist = sum(81 * (k - 1) + red + 9 for kw in sizes for k in range(2, kw))     # (`row_tables`: their constants as shifted tables)
        return (rf * t + rp) * chain + (rf + n_win) * t * row_finish + hist
    if optimised:
        # S-box layers: RF full, RP partial.  Linear layers: RF - 2 normalised dense (every full round but the entrance and the
        # last one) + 1 dense (last round) + RP sparse (after the entrance round and after every partial round but the last)
        # + 1 normalised dense (after the last partial round)
        return rf * t * chain + rp * chain + (rf - 2) * t * norm_dense + t * dot + rp * sparse + t * norm_dense
    return (rf * t + rp) * chain + (rf + rp) * t * dot


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default 200: a third of a second of kernel time at C2)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--total-log2", type=int, default=None, help="log2 of the units (states / leaves / rows) over ALL ranks")
    ap.add_argument("--total-units", type=int, default=None, help="units over ALL ranks, any number (ragged shards)")
    ap.add_argument("--states-per-gpu-log2", type=int, default=None, help="log2 of the units PER rank (weak scaling)")
    ap.add_argument("--gather", default="final", choices=["final", "step", "none", "root", "overlap", "overlap-root"],
                    help="N>1 permutation batches: 'final' = one RCCL all-gather of the result shards after the K steps (inside "
                         "the timed region); 'step' = an all-gather after EVERY step; 'root' = the shards go to rank 0 only (grouped "
                         "ncclSend / ncclRecv: 1/N of the bytes); 'overlap' / 'overlap-root' = the K-th step runs in --gather-chunks pieces "
                         "and piece i's transfers (to every rank / to rank 0) go behind piece i's kernel (pmx_mgpu_permute_gather_dev)")
    ap.add_argument("--gather-chunks", type=int, default=8, help="pieces of the last step in the overlap forms (1 .. 16)")
    ap.add_argument("--spinup-seconds", type=float, default=0.25, help="untimed device spin-up before the W warmup steps")
    ap.add_argument("--single-process", action="store_true",
                    help="N>1 without a launcher: ONE process drives all N GPUs through the C ABI's single-process device group "
                         "(pmx_mgpu_create = ncclCommInitAll; what a Rust caller uses) instead of starting one rank per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the all-cores cpu_baseline sample")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# the checker (oracle/): cpu_baseline and the post-run verification - never inside the timed region
# ----------------------------------------------------------------------------------------------------------------------
def oracle_engine(field_name, rate, alpha, rf, rp, handle=None):
    from oracle import cref
    from oracle import poseidon_oracle as O
    p, bits = {"bls12_381_fr": (O.BLS12_381_FR, 255), "bn254_fr": (O.BN254_FR, 254)}[field_name]
    return cref.CRef(O.make_config(p, bits, rate, alpha, rf, rp), handle)


def cpu_baseline(field_name, rate, alpha, rf, rp, seed, target_seconds):
    """C restatement of the reference permutation (oracle/poseidon_ref.c): on all host cores, and on ONE thread - the
    reference itself is single-threaded (src/poseidon/mod.rs:95-118)."""
    from oracle import cref
    import sponge_amd as S
    from sponge_amd import synth

    # BASELINE.md quotes the CPU path built with -march=native: compile the restatement for THIS host if gcc is here (the
    # build that travels with the repo is x86-64-v3, portable but without ADX), after checking it against the portable build
    native = cref.native_lib()
    cr = oracle_engine(field_name, rate, alpha, rf, rp, native)
    t = rate + 1
    threads = cref.max_threads()      # affinity mask capped by the cgroup CPU quota
    field = S.FIELDS[field_name]
    build = "-O3 -march=native, built on this host" if native else "-O3 -march=x86-64-v3 (portable build)"
    if native:
        probe = synth.random_elements(field, 64 * t, seed + 1).reshape(64, t, 4)
        if not np.array_equal(cr.permute_batch(probe, threads=1), oracle_engine(field_name, rate, alpha, rf, rp).permute_batch(probe, threads=1)):
            cr, build = oracle_engine(field_name, rate, alpha, rf, rp), "-O3 -march=x86-64-v3 (the native build disagreed with it: not used)"

    def timed(n_threads, seconds):
        probe = synth.random_elements(field, 1024 * t, seed).reshape(1024, t, 4)
        t0 = time.perf_counter()
        cr.permute_batch(probe, threads=n_threads)
        dt = max(time.perf_counter() - t0, 1e-6)
        n = int(min(1 << 22, max(1024, 1024 * seconds / dt)))
        n = 1 << (n.bit_length() - 1)
        batch = synth.random_elements(field, n * t, seed).reshape(n, t, 4)
        t0 = time.perf_counter()
        cr.permute_batch(batch, threads=n_threads)
        return n, time.perf_counter() - t0

    n, dt = timed(threads, target_seconds)
    n1, dt1 = timed(1, min(6.0, target_seconds / 2))
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    what = "C restatement of mod.rs:63-118 (dense MDS, square-and-multiply pow), gcc " + build
    return {"value": n / dt, "unit": "permutations/s", "cores": threads, "kind": "port",
            "sample": f"first {n} states of the same seeded batch, {what}, OpenMP x{threads}, {dt:.2f} s, CPU: {model}",
            "single_thread": {"value": n1 / dt1, "unit": "permutations/s", "cores": 1, "kind": "port",
                              "sample": f"first {n1} states of the same seeded batch, {what}, 1 thread (the reference's own "
                                        f"threading), {dt1:.2f} s"}}


def sample_indices(n, k):
    """k indices spread over [0, n): both ends and a stride in between."""
    if n <= k:
        return np.arange(n)
    return np.unique(np.concatenate([np.arange(0, min(n, k // 4)), np.arange(n - k // 4, n),
                                     np.linspace(0, n - 1, k // 2).astype(np.int64)]))


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher environment: start the N ranks OURSELVES, as a child process
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` - the very command the
    driver's contract spells), decided before torch is imported or any HIP call is made, never by exec; the child's exit
    code is ours and rank 0's JSON line passes through on stdout."""
    import socket
    import subprocess
    with socket.socket() as s:                 # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: launching the ranks as a child process: %s\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main_single_process(args):
    """`python bench.py --gpus N --single-process`: ONE process drives all N GPUs through the C ABI's single-process device group
    (pmx_mgpu_create = ncclCommInitAll - SURVEY section 7 step 6, and what a Rust caller's BatchPoseidon::new_multi does): no
    torch.distributed, no launcher.  torch only allocates the per-device buffers and the HIP events on the library's own streams.
    Same workloads, sizes, step / gather / verification semantics and JSON line as the one-rank-per-GPU form: contiguous shards, no
    collective on the data path, one final RCCL gather inside the timed region (permutation batches) or the roots all-gather of the
    sharded tree.  Time = host wall clock from the first enqueue to pmx_mgpu_synchronize returning, bracketed by synchronisation
    of every device on both sides; the kernel time is the longest per-device HIP-event interval.
    PMX_BENCH_REHEARSAL=group: the N slots share the visible GPU(s) (test-hook build of the library + a collective library that
    accepts that: tests/fake_rccl first on LD_LIBRARY_PATH or named by PMX_RCCL_LIBRARY) - every branch executes, no number means anything."""
    import torch
    import sponge_amd as S
    from sponge_amd import _lib, mgpu, synth

    world = args.gpus
    mode = os.environ.get("PMX_BENCH_REHEARSAL", "")
    if mode not in ("", "group"):
        raise SystemExit("--single-process: PMX_BENCH_REHEARSAL is unset or 'group'")
    rehearsal = mode == "group"
    if rehearsal:
        _lib.use_test_library()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Poseidon path has no CPU fallback")
    if args.workload in HASH_SHAPES or args.workload in DUPLEX_SHAPES:
        raise SystemExit("--single-process runs the permutation batches (c2, c3, k3, w*) and the sharded tree (c5)")
    visible = torch.cuda.device_count()
    if rehearsal:
        devices = [l % visible for l in range(world)]
        _lib.check(_lib.lib().pmx_mgpu_test_shared_device(1))
    else:
        if world > visible:
            raise SystemExit(f"--gpus {world} but only {visible} visible device(s)")
        devices = list(range(world))

    field_name, rate, alpha, rf, rp, log2_one, log2_total_multi, seed, desc = WORKLOADS[args.workload]
    t = rate + 1
    merkle = args.workload == "c5"
    if args.total_units is not None:
        n_total, scaling, baseline_cfg = args.total_units, "strong", None
    elif args.total_log2 is not None:
        n_total, scaling, baseline_cfg = 1 << args.total_log2, "strong", None
    elif args.states_per_gpu_log2 is not None:
        n_total, scaling, baseline_cfg = world << args.states_per_gpu_log2, "weak", None
    elif log2_total_multi is not None:
        n_total, scaling, baseline_cfg = 1 << log2_total_multi, "strong", BASELINE_CONFIG.get((args.workload, True))
    else:
        n_total, scaling, baseline_cfg = world << log2_one, "weak", None
    if merkle and (n_total % world or (n_total // world) & (n_total // world - 1) or world & (world - 1)):
        raise SystemExit("the Merkle workload needs power-of-two leaves and ranks")
    field = S.FIELDS[field_name]
    cfg = S.poseidon_config_from_lfsr(field, rate, alpha, rf, rp)

    peak = _lib.PmxValuPeak()
    _lib.check_diag(_lib.diag_lib().pmx_diag_int_valu_peak(devices[0], 0.1, peak))
    group = mgpu.DeviceGroup.single_process(cfg, devices=devices)
    info = group.info()
    rccl = {"ranks": info["comm_ranks"], "version": info["rccl_version_str"], "rank0_is": info["comm_first_rank"],
            "via": "pmx_mgpu_create (ncclCommInitAll, one process); gather = " + gather_entry(args)}
    if rehearsal:
        rccl["via"] += "; REHEARSAL: the slots share GPUs, collective library = " + os.environ.get("PMX_RCCL_LIBRARY", "librccl.so.1 on the search path")
    ctx = group.context(0)
    spans = [group.local_span(n_total, l) for l in range(world)]
    n = spans[0][1]
    slot = {}
    early = _lib.PmxEngineInfo()
    _lib.check(_lib.lib().pmx_ctx_engine_info(ctx._h, _lib.OP_COMPRESS if merkle else _lib.OP_PERMUTE, (n // 2 if merkle else n), 0, early))
    r = _lib.PmxIssueSlot()
    _lib.check_diag(_lib.diag_lib().pmx_diag_issue_slot(devices[0], max(1, min(8, early.waves_per_simd)), 0.06, r))
    slot[max(1, min(8, early.waves_per_simd))] = r
    devs = [torch.device("cuda", d) for d in devices]
    streams = [torch.cuda.ExternalStream(group.stream(l), device=devs[l]) for l in range(world)]

    def sync_all():
        group.synchronize()
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    def fresh_inputs(l):
        start, count = spans[l]
        host = synth.random_elements(field, count * (1 if merkle else t), seed, offset=start * (1 if merkle else t))
        return host, torch.from_numpy(host.view(np.int64).copy()).to(devs[l])

    def make_buffers():
        b = {"in": [], "top": [], "gathered": [], "host": []}
        for l in range(world):
            host, d_in = fresh_inputs(l)
            count = spans[l][1]
            b["host"].append(host)
            if merkle:
                nodes = torch.zeros((2 * count - 1, 4), dtype=torch.int64, device=devs[l])
                nodes[:count] = d_in.reshape(count, 4)
                b["in"].append(nodes)
                b["top"].append(torch.zeros((2 * world - 1, 4), dtype=torch.int64, device=devs[l]))
            else:
                b["in"].append(d_in.reshape(count, t, 4))
                if args.gather != "none":      # (to rank 0 only: the other slots have no buffer for the result)
                    b["gathered"].append(torch.empty((n_total, t, 4), dtype=torch.int64, device=devs[l]) if (l == 0 or not to_root) else None)
        return b

    to_root, overlapped = args.gather in ("root", "overlap-root"), args.gather in ("overlap", "overlap-root")

    def gathered_ptrs(b):
        return [x.data_ptr() if x is not None else 0 for x in b["gathered"]]

    def run_step(b, last=False):
        if merkle:
            group.merkle_2to1_dev([x.data_ptr() for x in b["in"]], [x.data_ptr() for x in b["top"]], n_total)
        elif last and overlapped:                                                  # the K-th step and its gather, piece by piece
            group.permute_gather_dev([x.data_ptr() for x in b["in"]], gathered_ptrs(b), n_total, 0 if to_root else -1, args.gather_chunks)
        else:
            group.permute_shards_dev([x.data_ptr() for x in b["in"]], n_total)       # no collective on the data path
            if args.gather == "step":
                run_gather(b)

    def run_gather(b):
        if to_root:
            group.gather_dev([x.data_ptr() for x in b["in"]], gathered_ptrs(b), n_total, t, 0)
        else:
            group.all_gather_dev([x.data_ptr() for x in b["in"]], gathered_ptrs(b), n_total, t)

    def final_gather(b):
        if not merkle and args.gather in ("final", "root"):
            run_gather(b)

    bufs = make_buffers()
    units_per_step = float(n_total - 1) if merkle else float(n_total)
    sync_all()
    t_spin, i_spin = time.perf_counter(), 0
    while (i_spin < 32) if merkle else (time.perf_counter() - t_spin < args.spinup_seconds):
        run_step(bufs)
        i_spin += 1
        if i_spin % 4 == 0:
            group.synchronize()
    for _ in range(args.warmup):
        run_step(bufs)
    final_gather(bufs)          # also warms RCCL's lazily built channels up, outside the timed region
    if overlapped and not merkle:
        run_step(bufs, last=True)
    sync_all()

    def events():
        out = []
        for l in range(world):
            with torch.cuda.device(devs[l]):
                e = torch.cuda.Event(enable_timing=True)
                e.record(streams[l])
                out.append(e)
        return out

    t0 = time.perf_counter()
    ev0 = events()
    for i in range(args.steps):
        run_step(bufs, last=i + 1 == args.steps)
    ev_k = events()
    final_gather(bufs)
    ev1 = events()
    sync_all()
    elapsed = time.perf_counter() - t0
    dev_s = max(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / 1e3
    steps_s = max(a.elapsed_time(b) for a, b in zip(ev0, ev_k)) / 1e3

    verify = None
    if not args.no_verify:
        cr = oracle_engine(field_name, rate, alpha, rf, rp)
        verify = {"ok": True, "engine": "oracle/poseidon_ref.c (C restatement)", "what": None}
        b = make_buffers()
        for g in b["gathered"]:
            if g is not None:
                g.zero_()
        sync_all()
        run_step(b, last=True)
        if not merkle and b["gathered"] and args.gather in ("final", "root"):
            run_gather(b)
        sync_all()
        ok, checked = True, 0

        def to_np(x):
            return x.cpu().numpy().view(np.uint64)
        if merkle:
            tops = [to_np(x) for x in b["top"]]
            for l in range(world):
                count = spans[l][1]
                got = to_np(b["in"][l])
                if count >= 2:
                    idx = sample_indices(count // 2, 1024)
                    pairs = np.ascontiguousarray(b["host"][l].reshape(count // 2, 2, 4)[idx])
                    ok &= bool(np.array_equal(got[count + idx], cr.hash_batch(pairs, 2, 1, threads=0).reshape(-1, 4)))
                w = min(count, 1024)
                first = 2 * count - 2 * w
                ok &= bool(np.array_equal(cr.merkle(np.ascontiguousarray(got[first:first + w]), threads=0)[w:], got[first + w:]))
                ok &= bool(np.array_equal(tops[l][l], got[-1]))                                               # slot l's root at rank l
                ok &= bool(np.array_equal(cr.merkle(np.ascontiguousarray(tops[l][:world]), threads=0), tops[l]))   # every device holds the whole top
            verify["what"] = f"a fresh tree on every device: level 1 on a sample, the top of each subtree, the {world} gathered roots and the levels above them on every device"
        elif b["gathered"]:
            for l in range(world):                  # every device's copy of the gathered result, a sample of every rank's span
                if b["gathered"][l] is None:
                    continue
                for rr in range(world):
                    s_r, c_r = spans[rr]
                    idx = s_r + sample_indices(c_r, 128)
                    inp = np.stack([synth.random_elements(field, t, seed, offset=int(i) * t) for i in idx])
                    ok &= bool(np.array_equal(to_np(b["gathered"][l][torch.from_numpy(idx).to(devs[l])]), cr.permute_batch(inp, threads=0)))
                    checked += len(idx)
            verify["what"] = (f"the gathered buffer of one fresh pass on {'rank 0' if to_root else 'every device'}: {checked} states, a sample of every rank's span at its offset"
                              + (f" (the pass: pmx_mgpu_permute_gather_dev in {args.gather_chunks} pieces)" if overlapped else ""))
        else:
            for l in range(world):
                count = spans[l][1]
                idx = sample_indices(count, 1024)
                want = cr.permute_batch(np.ascontiguousarray(b["host"][l].reshape(count, t, 4)[idx]), threads=0)
                ok &= bool(np.array_equal(to_np(b["in"][l])[idx], want))
            verify["what"] = "a sample of every device's shard of one fresh pass (--gather none)"
        verify["ok"] = ok

    out = result_line(args, ctx, peak, slot, world=world, n=n, n_total=n_total, units_per_step=units_per_step, elapsed=elapsed,
                      steps_s=steps_s, dev_s=dev_s, scaling=scaling, baseline_cfg=baseline_cfg, rccl=rccl, verify=verify, rehearsal=rehearsal,
                      launcher="one process, every GPU through pmx_mgpu_create (ncclCommInitAll); no torch.distributed")
    out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    group.close()
    if verify is not None and not verify["ok"]:
        sys.exit(3)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: either one process drives every GPU through the library's own device group
        # (--single-process: pmx_mgpu_create = ncclCommInitAll, no torch.distributed at all), or - the default, the form the
        # contract names - we start one rank per GPU as a child.  Decided here, before torch is imported.
        if args.single_process:
            return main_single_process(args)
        sys.exit(launch_ranks(args))
    if args.single_process and args.gpus > 1:
        raise SystemExit("--single-process drives every GPU from ONE process: start it without a launcher (WORLD_SIZE is set)")
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Poseidon path has no CPU fallback")

    import torch.distributed as dist
    import sponge_amd as S
    from sponge_amd import _lib, mgpu, synth

    # PMX_BENCH_REHEARSAL=group: a dry run of the N > 1 code path on a box with ONE GPU, through the PRODUCT's multi-rank path - the C
    # ABI's device group (pmx_mgpu_create_rank, pmx_mgpu_permute_shards_dev, pmx_mgpu_all_gather_dev, pmx_mgpu_merkle_2to1_dev) exactly as
    # in a real N > 1 run; the ranks share the visible GPUs round-robin and only torch.distributed's control plane is on gloo.  The
    # collective library then has to accept several ranks on one device, which RCCL does not: the caller names one that does with
    # PMX_RCCL_LIBRARY (tools/gpu_group_rehearsal.sh uses the tests' stand-in).  Every branch executes, no number means anything, and
    # the line says so.  (There is no other gather path: a run that cannot form the device group fails.)
    rehearsal_mode = os.environ.get("PMX_BENCH_REHEARSAL", "")
    if rehearsal_mode not in ("", "group"):
        raise SystemExit("PMX_BENCH_REHEARSAL is unset or 'group' (the C ABI's device group on shared GPUs)")
    rehearsal = rehearsal_mode != ""
    if rehearsal:
        _lib.use_test_library()      # only the test-hook build reads PMX_RCCL_LIBRARY (the stand-in that accepts ranks sharing a GPU)
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ctl = torch.device("cpu") if rehearsal else dev      # where the control-plane tensors of torch.distributed live

    field_name, rate, alpha, rf, rp, log2_one, log2_total_multi, seed, desc = WORKLOADS[args.workload]
    t = rate + 1
    merkle = args.workload == "c5"
    hashing = args.workload in HASH_SHAPES
    duplex = args.workload in DUPLEX_SHAPES
    if duplex and (world > 1 or args.warmup < 1):
        raise SystemExit("the duplex-driver workloads run on one GPU and need --warmup >= 1 (the first step takes the fresh sponges into their steady cycle)")
    # ---- sizes: the BASELINE configuration of this (workload, N), unless overridden -----------------------------------
    if args.total_units is not None:
        n_total, scaling, baseline_cfg = args.total_units, "strong", None
    elif args.total_log2 is not None:
        n_total, scaling, baseline_cfg = 1 << args.total_log2, "strong", None
    elif args.states_per_gpu_log2 is not None:
        n_total, scaling, baseline_cfg = world << args.states_per_gpu_log2, "weak", None
    elif world > 1 and log2_total_multi is not None:
        n_total, scaling = 1 << log2_total_multi, "strong"
        baseline_cfg = BASELINE_CONFIG.get((args.workload, True))
    else:
        n_total, scaling = world << log2_one, ("strong" if world == 1 else "weak")
        baseline_cfg = BASELINE_CONFIG.get((args.workload, False)) if world == 1 else None
    start, n = mgpu.shard_bounds(n_total, world, rank)       # this rank's contiguous shard
    if merkle and (n_total % world or n & (n - 1) or world & (world - 1)):
        raise SystemExit("the Merkle workload needs power-of-two leaves and ranks")
    field = S.FIELDS[field_name]
    cfg = S.poseidon_config_from_lfsr(field, rate, alpha, rf, rp)

    # ---- per-run integer-VALU roofline of THIS device, before anything is timed -----------------------------------------
    peak = _lib.PmxValuPeak()
    _lib.check_diag(_lib.diag_lib().pmx_diag_int_valu_peak(local_rank, 0.1, peak))
    slot = {}          # waves per SIMD -> in-run issue-slot measurement (pmx_diag_issue_slot), filled for the kernel's occupancy below

    def issue_slot(waves):
        if waves not in slot:
            r = _lib.PmxIssueSlot()
            _lib.check_diag(_lib.diag_lib().pmx_diag_issue_slot(local_rank, waves, 0.06, r))
            slot[waves] = r
        return slot[waves]

    # ---- the engine: one context at N = 1, the C ABI's device group (RCCL) at N > 1 -------------------------------------
    group, group_error, rccl = None, None, None
    if world > 1:
        uid = [mgpu.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0, device=ctl)
        try:
            group = mgpu.DeviceGroup.one_rank(cfg, local_rank, rank, world, uid[0])
            info = group.info()
            rccl = {"ranks": info["comm_ranks"], "version": info["rccl_version_str"], "rank0_is": info["comm_first_rank"],
                    "via": "pmx_mgpu_create_rank (ncclCommInitRank); gather = " + gather_entry(args)}
        except S.PmxError as e:
            group_error = str(e)
        if group is not None and rehearsal:
            rccl["via"] += "; REHEARSAL: the ranks share GPUs, collective library = " + os.environ.get("PMX_RCCL_LIBRARY", "librccl.so.1")
        ok = torch.tensor([1 if group is not None else 0], device=ctl)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok) == 0:
            # The product's gather is pmx_mgpu_all_gather_dev.  A run that cannot form the device group is a failed run, not a
            # run on another code path.
            if group is not None:      # all ranks leave together
                group.close()
                group = None
            sys.stderr.write("rank %d: the C ABI's device group could not be formed on every rank (%s)\n" % (rank, group_error or "another rank failed"))
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(4)
    ctx = cfg.context(local_rank)
    if rank == 0:       # the issue slot at the dominant kernel's occupancy, measured before anything is timed
        early = _lib.PmxEngineInfo()
        _lib.check(_lib.lib().pmx_ctx_engine_info(ctx._h, _lib.OP_COMPRESS if args.workload == "c5" else (_lib.OP_HASH if args.workload in HASH_SHAPES else
                                                  (_lib.OP_ABSORB if args.workload in DUPLEX_SHAPES else _lib.OP_PERMUTE)),
                                                  (n // 2 if args.workload == "c5" else n), DUPLEX_SHAPES.get(args.workload, (0,))[0], early))
        issue_slot(max(1, min(8, early.waves_per_simd)))
    if group is not None:
        stream = torch.cuda.ExternalStream(group.stream(0), device=dev)     # the library's stream of this device
    else:
        stream = torch.cuda.current_stream()

    def fresh_inputs():
        """this rank's shard of the global seeded input, uploaded (used for the timed buffers and again for the check)"""
        if hashing or duplex:
            host = synth.random_elements(field, n * in_len, seed, offset=start * in_len)
        elif merkle:
            host = synth.random_elements(field, n, seed, offset=start)
        else:
            host = synth.random_elements(field, n * t, seed, offset=start * t)
        return host, torch.from_numpy(host.view(np.int64).copy()).to(dev)

    if duplex:
        in_len, out_len, perms_per_row = DUPLEX_SHAPES[args.workload]
        host_in, d_in = fresh_inputs()
        d_out = torch.zeros((n, out_len, 4), dtype=torch.int64, device=dev)
        d_st = torch.zeros((n, t, 4), dtype=torch.int64, device=dev)        # CryptographicSponge::new: zero state, Absorbing{0}
        d_tag = torch.zeros(n, dtype=torch.int32, device=dev)
        d_idx = torch.zeros(n, dtype=torch.int32, device=dev)
        units_per_step = float(n_total) * perms_per_row

        def step(i):
            ctx.sponge_absorb_batch_dev(d_st.data_ptr(), d_tag.data_ptr(), d_idx.data_ptr(), d_in.data_ptr(), in_len, n, stream.cuda_stream)
            ctx.sponge_squeeze_batch_dev(d_st.data_ptr(), d_tag.data_ptr(), d_idx.data_ptr(), d_out.data_ptr(), out_len, n, stream.cuda_stream)

        def final_gather():
            pass
    elif hashing:
        in_len, out_len, perms_per_row = HASH_SHAPES[args.workload]
        host_in, d_in = fresh_inputs()
        d_out = torch.zeros((n, out_len, 4), dtype=torch.int64, device=dev)
        units_per_step = float(n_total) * perms_per_row

        def step(i):
            ctx.hash_batch_dev(d_in.data_ptr(), in_len, d_out.data_ptr(), out_len, n, stream.cuda_stream)

        def final_gather():
            pass
    elif merkle:
        # leaves of this rank's subtree, resident in the first n rows of the node array [2n-1][4]
        host_in, d_leaves = fresh_inputs()
        nodes = torch.zeros((2 * n - 1, 4), dtype=torch.int64, device=dev)
        nodes[:n] = d_leaves.reshape(n, 4)
        top = torch.zeros((max(2 * world - 1, 1), 4), dtype=torch.int64, device=dev)
        units_per_step = float(n_total - 1)                             # compressions per step, all ranks

        def step(i):
            if group is not None:
                group.merkle_2to1_dev([nodes.data_ptr()], [top.data_ptr()], n_total)
            else:
                ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)

        def final_gather():
            pass
    else:
        host_in, buf = fresh_inputs()
        buf = buf.reshape(n, t, 4)
        to_root, overlapped = args.gather in ("root", "overlap-root"), args.gather in ("overlap", "overlap-root")
        receives = not to_root or rank == 0          # (to rank 0 only: the other ranks have no buffer for the result)
        gathered = torch.empty((n_total, t, 4), dtype=torch.int64, device=dev) if (world > 1 and args.gather != "none" and receives) else None
        units_per_step = float(n_total)

        def gather_now(src):
            dst = [gathered.data_ptr() if gathered is not None else 0]
            if to_root:
                group.gather_dev([src.data_ptr()], dst, n_total, t, 0)
            else:
                group.all_gather_dev([src.data_ptr()], dst, n_total, t)      # (world > 1: the device group exists)

        def step(i, last=False):
            if group is not None and last and overlapped:                  # the K-th step and its gather, piece by piece
                group.permute_gather_dev([buf.data_ptr()], [gathered.data_ptr() if gathered is not None else 0], n_total, 0 if to_root else -1, args.gather_chunks)
            elif group is not None:
                group.permute_shards_dev([buf.data_ptr()], n_total)        # no collective on the data path
            else:
                ctx.permute_batch_dev(buf.data_ptr(), n, stream.cuda_stream)
            if world > 1 and args.gather == "step":
                gather_now(buf)

        def final_gather():                      # the job's epilogue: every rank (or rank 0) ends with the whole result
            if world > 1 and args.gather in ("final", "root"):
                gather_now(buf)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.cuda.synchronize()
    # Device spin-up (untimed, not counted as steps): the shader clock ramps over the first ~25 ms of activity
    # (per-launch time falls from 2.5 ms to the steady value across the first dozen launches), so the W warmup steps alone
    # would leave short runs measuring the ramp instead of the kernel.  A step that contains a collective must run the
    # same number of times on every rank: fixed count instead of a clock.
    collective_in_step = world > 1 and (merkle or args.gather == "step")
    t_spin, i_spin = time.perf_counter(), 0
    while (i_spin < 32) if collective_in_step else (time.perf_counter() - t_spin < args.spinup_seconds):
        step(i_spin)
        i_spin += 1
        if i_spin % 4 == 0:
            torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    final_gather()      # also warms RCCL's lazily built rings up, outside the timed region
    last_step_gathers = world > 1 and not (duplex or hashing or merkle) and args.gather in ("overlap", "overlap-root")
    if last_step_gathers:
        step(0, last=True)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        if last_step_gathers and i + 1 == args.steps:
            step(i, last=True)
        else:
            step(i)
    ev_k = torch.cuda.Event(enable_timing=True)
    ev_k.record(stream)
    final_gather()
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)          # HIP events on the launch stream
    steps_ms = ev0.elapsed_time(ev_k)       # ... the K steps without the epilogue gather

    times = torch.tensor([elapsed, dev_ms / 1e3, steps_ms / 1e3], dtype=torch.float64, device=ctl)
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    elapsed, dev_s, steps_s = float(times[0]), float(times[1]), float(times[2])

    # ---- verification (untimed): one pass over a FRESH copy of the seeded inputs, sample against the C restatement ------
    verify = None
    if not args.no_verify:
        verify = run_verification(locals())
        flag = torch.tensor([1 if verify["ok"] else 0], device=ctl)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verify["ok"] = bool(int(flag))

    if rank == 0:
        out = result_line(args, ctx, peak, slot, world=world, n=n, n_total=n_total, units_per_step=units_per_step, elapsed=elapsed,
                          steps_s=steps_s, dev_s=dev_s, scaling=scaling, baseline_cfg=baseline_cfg, rccl=rccl, verify=verify, rehearsal=rehearsal,
                          in_len=(in_len if (hashing or duplex) else 0), out_len=(out_len if (hashing or duplex) else 0),
                          perms_per_row=(perms_per_row if (hashing or duplex) else 1))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(field_name, rate, alpha, rf, rp, seed, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        if group is not None:
            group.close()
        dist.destroy_process_group()
    if verify is not None and not verify["ok"]:
        sys.exit(3)


def result_line(args, ctx, peak, slot, *, world, n, n_total, units_per_step, elapsed, steps_s, dev_s, scaling, baseline_cfg, rccl, verify,
                rehearsal, in_len=0, out_len=0, perms_per_row=1, launcher=None):
    """The JSON line (everything but cpu_baseline) from a run's sizes and times: one function for the one-rank-per-GPU form
    and the single-process form, so that the N = 1 line and both N > 1 lines price the kernels the same way."""
    from sponge_amd import _lib
    field_name, rate, alpha, rf, rp, _l1, _lm, _seed, desc = WORKLOADS[args.workload]
    t = rate + 1
    merkle = args.workload == "c5"
    hashing = args.workload in HASH_SHAPES
    duplex = args.workload in DUPLEX_SHAPES
    value = units_per_step * args.steps / elapsed
    # dominant kernel: permute_kernel (compress kernels in the Merkle mode); its average launch duration from the HIP
    # events on this rank's launch stream around the K steps (back-to-back launches, nothing else on the stream)
    kernel_s = steps_s / args.steps
    per_gpu_units = units_per_step / world
    bytes_per_unit = 96 if merkle else 2 * t * 32       # 2*t*32 B per permutation; 64 in + 32 out per 2-to-1
    if hashing:
        bytes_per_unit = (in_len + out_len) * 32 / perms_per_row
    if duplex:      # per sponge and step: the input row, the output row, and every permutation's state once in and once out
        bytes_per_unit = ((in_len + out_len) * 32 + perms_per_row * 2 * t * 32) / perms_per_row
    algo_bytes = bytes_per_unit * per_gpu_units
    achieved = algo_bytes / kernel_s / 1e9
    # The engine as the library's own launchers dispatch it for this call (pmx_ctx_engine_info: schedule, table forms,
    # matrix-core rows, launch bound) - the instruction accounting below follows the kernels, not a copy of their rules.
    # ABI <-> internal conversions cost no multiplies on the optimised schedule (pmx_field.hpp: fe_from_abi_scaled).
    info = _lib.PmxEngineInfo()
    op = _lib.OP_COMPRESS if merkle else (_lib.OP_HASH if hashing else (_lib.OP_ABSORB if duplex else _lib.OP_PERMUTE))
    _lib.check(_lib.lib().pmx_ctx_engine_info(ctx._h, op, (n // 2 if merkle else n), (in_len if duplex else 0), info))
    mfma_dense = bool(info.mfma_dense)
    mads = mads_per_permutation(t, alpha, rf, rp, optimised=bool(info.optimised), row_tables=int(info.row_tables),
                                lane_tables=bool(info.lane_tables), mfma_dense=mfma_dense, window=int(info.partial_window))
    # the last round of a permutation whose caller reads only some lanes computes only those rows (pmx_permute.hpp:
    # want_lo / want_hi): the digest lane of a 2-to-1 compression, the out_len lanes of a hash row's last permutation
    last_row = 8 if mfma_dense else 81 * t + (18 if info.row_tables == 1 else 81)   # (8: the row finish of mads_per_permutation)
    if merkle:
        mads -= (t - 1) * last_row
    elif hashing:
        mads -= (t - out_len) * last_row / perms_per_row
    mad_rate = mads * per_gpu_units / kernel_s
    traffic, traffic_src = load_traffic(args.workload, per_gpu_units)
    valu_issue = load_valu_issue(args.workload, per_gpu_units, kernel_s, peak.compute_units, mads, info, slot)
    if mfma_dense and not (merkle or hashing or duplex):
        price_products(valu_issue, matrix_products_per_permutation(t, rf, rp, int(info.partial_window), int(info.row_tables) == 2), peak.shader_clock_hz)
    out = {
        "metric": "Poseidon permutations/sec (%s, t=%d)" % ({"bls12_381_fr": "BLS12-381 Fr", "bn254_fr": "BN254 Fr"}[field_name], t),
        "value": value, "unit": "permutations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"{desc}, {fmt_pow2(n_total)} units in all = {fmt_pow2(n)} per GPU x {world}",
                   "baseline_config": baseline_cfg,
                   "arithmetic": "255-bit modular integers as 9 x 29-bit limbs in u32, Montgomery form",
                   "units_total": n_total, "units_per_gpu": n, "permutations_per_step": units_per_step,
                   "gather": (args.gather if (world > 1 and not merkle and not hashing) else ("roots" if merkle and world > 1 else "n/a")),
                   "sharding": f"contiguous x{world}", **({"REHEARSAL": "ranks share one GPU, gloo: not a measurement"} if rehearsal else {}),
                   "series": "N=1 runs configs[1] (2^20 states); N>1 shard configs[3]'s 2^24 states (strong scaling)"
                             if args.workload == "c2" and baseline_cfg else None},
        "engine": {"name": info.engine.decode(), "threads_per_workgroup": info.threads, "waves_per_simd": info.waves_per_simd,
                   "lds_bytes_per_workgroup": info.lds_bytes, "optimised_schedule": bool(info.optimised),
                   "row_tables": bool(info.row_tables), "history_rows_on_matrix_cores": int(info.row_tables) == 2, "lane_tables": bool(info.lane_tables), "mfma_dense": mfma_dense,
                   "partial_window": int(info.partial_window),
                   "source": "pmx_ctx_engine_info (the launchers' own dispatch conditions)" + (", widest level of the tree" if merkle else "")},
        "rccl": rccl,
        "verified": (verify["ok"] if verify else None), "verify": verify,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "pmx::compress_kernel / compress_coop_kernel (per tree level)" if merkle else
                               ("pmx::hash_kernel" if hashing else ("the driver's kernels: " + info.engine.decode() if duplex else "pmx::permute_kernel")),
                     "kernel_ms": 1e3 * kernel_s, "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "integer-VALU bound, not HBM bound (DESIGN.md): see int_valu"},
        "int_valu": {"bound": "v_mad_u64_u32 issue", "achieved": mad_rate, "peak": peak.lane_mads_per_s,
                     "unit": "lane-instr/s", "frac": mad_rate / peak.lane_mads_per_s, "mads_per_permutation": mads,
                     "dense_layers": "v_mfma_i32_32x32x32_i8 (pmx_mfma.hpp): not counted as multiplies" if mfma_dense else "VALU",
                     "model_note": ("an UPPER bound of the multiplies a permutation executes: pmx_ctx_engine_info does not say which t = 3 windows take their history "
                                    "term as additions instead of a table product (all ten of BASELINE's C2 / C5 config do: 108 multiplies each) nor that the first "
                                    "permutation of a compression / hash row takes lane 0's round-0 S-box from a constant - this rate and `frac` are up to 5 % high "
                                    "at t = 3; valu_issue (a counter, not a model) is the figure that binds"),
                     "peak_source": "pmx_diag_int_valu_peak on this device just before the warm-up (%d launches of a dense v_mad_u64_u32 loop, two forms, median of the later half of each)" % peak.launches,
                     "peak_best_launch": peak.best_lane_mads_per_s,
                     "peak_by_carry_destination": {"vcc": peak.lane_mads_per_s_vcc, "sgpr_pair": peak.lane_mads_per_s_sgpr}, "shader_clock_hz": peak.shader_clock_hz,
                     "theoretical_peak": peak.theoretical_lane_mads_per_s, "compute_units": peak.compute_units,
                     "frac_of_theoretical": mad_rate / peak.theoretical_lane_mads_per_s if peak.theoretical_lane_mads_per_s else None},
        "valu_issue": valu_issue,
        # the same run in units that do not move with the clock a box happens to hold (boxes of this pool hold 2.11 ... 2.25 GHz under this load):
        # shader clocks one SIMD spends per permutation, and - where the instruction count is known - clocks per VALU instruction and SIMD
        # (the issue slot is 4: one wave instruction per 16 lanes x 4 passes)
        "clock_normalised": clock_normalised(kernel_s, peak.shader_clock_hz, peak.compute_units, units_per_step / max(world, 1), valu_issue),
        # (overlap forms: the transfers run inside the K-th step - there is no epilogue to time; compare ms_per_step x steps with --gather none)
        "gather_ms": 1e3 * (dev_s - steps_s) if (world > 1 and args.gather not in ("overlap", "overlap-root")) else None,
        "gather_inside_last_step": (world > 1 and args.gather in ("overlap", "overlap-root") and not (merkle or hashing or duplex)) or None,
        # what the final gather should take if every peer's shard arrives over its own xGMI link at the link's rate (MI355X: 7 links x
        # ~153 GB/s per GPU, point to point): one shard's bytes / 153 GB/s, for up to 8 GPUs of one node - the first multi-GPU run judges
        # itself against this (tools/first_8gpu_run.sh prints both)
        "gather_ms_predicted": gather_model(args, world, n, t, merkle or hashing or duplex)[0],
        "gather_model": gather_model(args, world, n, t, merkle or hashing or duplex)[1],
    }
    if launcher:
        out["config"]["launcher"] = launcher
    return out


def gather_entry(args):
    """the C ABI entry point (and RCCL calls) behind --gather"""
    return {"root": "pmx_mgpu_gather_dev (grouped ncclSend / ncclRecv)",
            "overlap": "pmx_mgpu_permute_gather_dev, to every rank (grouped ncclSend / ncclRecv, piece by piece)",
            "overlap-root": "pmx_mgpu_permute_gather_dev, to rank 0 (grouped ncclSend / ncclRecv, piece by piece)",
            "none": "none"}.get(args.gather, "pmx_mgpu_all_gather_dev (ncclAllGather / grouped ncclBroadcast)")


def gather_model(args, world, n, t, not_a_permutation_batch):
    """(milliseconds, text): what the gather should ADD to the K steps if every shard travels over its own xGMI link at the link's rate
    (MI355X: 7 links x ~153 GB/s per GPU and direction, point to point).  To every rank or to rank 0 alike, a receiver takes world - 1
    shards over world - 1 links at once - one shard's time; the root form moves 1 / world of the bytes in all and leaves the other
    ranks' links idle.  Overlapped with the last step, only the last piece's transfer is left behind the kernels."""
    if world <= 1 or not_a_permutation_batch or args.gather == "none":
        return None, None
    shard = n * t * 32
    one = 1e3 * shard / 153e9
    if args.gather in ("final", "step"):
        return one, ("one ncclAllGather of %d shards of %d bytes; every GPU receives %d of them, each over its own xGMI link (153 GB/s per link and direction)"
                     % (world, shard, world - 1))
    if args.gather == "root":
        return one, ("grouped ncclSend / ncclRecv: %d shards of %d bytes to rank 0, each over its own xGMI link (153 GB/s); 1/%d of the all-gather's bytes in all"
                     % (world - 1, shard, world))
    return one / args.gather_chunks, ("the last step in %d pieces, piece i's transfers (%s) behind piece i's kernel on a second stream: one piece's transfer (%d bytes per "
                                       "link) is left behind the last kernel" % (args.gather_chunks, "to rank 0" if args.gather == "overlap-root" else "to every rank",
                                                                                shard // args.gather_chunks))


def matrix_products_per_permutation(t, rf, rp, window, hist_rows):
    """v_mfma_i32_32x32x32_i8 instructions one wave issues per permutation batch of 64 (pmx_permute.hpp: permute_hybrid): two per k-step
    (states 0-31 and 32-63 of the wave), one k-step per input element of a row.  Every full round closes with a layer of t rows over t
    inputs; the partial rounds run as ceil(rp / K) windows - the first the short one - each closed by a layer of t rows over t - 1 + K
    inputs, and (t >= 4) the S-box inputs x_3 .. x_K of a window are rows of 2 .. K - 1 inputs.  (C3: 1296 + 2142 + 424 = 3862, the
    count of the SQ_INSTS_VALU_MFMA pass in profiles/r06/z_2.24GHz_pmc_c3_stalls.txt.)"""
    products = rf * t * 2 * t
    if window and rp:
        n_win = -(-rp // window)
        first = rp - (n_win - 1) * window
        products += n_win * t * 2 * (t - 1 + window)
        if hist_rows:
            for kw in [first] + [window] * (n_win - 1):
                products += sum(2 * k for k in range(2, window) if k < kw)
    return products


def price_products(valu_issue, products, clock_hz):
    """The floor of `valu_issue` prices VALU instructions only.  A matrix-core product is not free on that port: spread evenly through
    a wave's multiplies it takes 11.6 clocks of it, issued as a layer issues them - a burst of 2 x n_in per row from two free-running
    waves per SIMD - 19.5 (tools/mfma_burst_microbench.hip -> profiles/r06/d_mfma_burst_microbench.txt; constants of the part, measured
    once, scaled by this run's clock).  `frac_products_priced`: (instructions x slot + products x price) / measured time, at both prices."""
    per, ns, floor = (valu_issue.get(k) for k in ("valu_instructions_per_permutation", "ns_per_instruction_and_simd", "floor_ns_per_instruction_and_simd"))
    valu_issue["matrix_products_per_permutation"] = products
    if not (per and ns and floor and clock_hz):
        return
    priced = {}
    for name, clocks in (("spread through the VALU work (11.6 clocks of the port)", 11.6), ("in a layer's bursts (19.5 clocks of the port)", 19.5)):
        priced[name] = (per * floor + products * clocks / clock_hz * 1e9) / (per * ns)
    valu_issue["frac_products_priced"] = priced
    valu_issue["products_price_source"] = "profiles/r06/d_mfma_burst_microbench.txt (tools/mfma_burst_microbench.hip), clocks x this run's shader clock"


def clock_normalised(kernel_s, clock_hz, compute_units, permutations_per_gpu_step, valu_issue):
    if not clock_hz or not compute_units or not permutations_per_gpu_step:
        return None
    simd_clocks = kernel_s * clock_hz * compute_units * 4          # SIMD-clocks the step's kernels had, over the whole chip
    out = {"shader_clock_hz": clock_hz, "clock_source": "pmx_diag_int_valu_peak's s_memtime / wall-clock ratio under a dense multiply stream, just before the warm-up",
           "simd_clocks_per_permutation": simd_clocks / permutations_per_gpu_step,
           "permutations_per_s_at_2.2GHz": permutations_per_gpu_step / kernel_s * (2.2e9 / clock_hz), "kernel_ms_at_2.2GHz": 1e3 * kernel_s * clock_hz / 2.2e9}
    per = (valu_issue or {}).get("valu_instructions_per_permutation")
    if per:
        # (a wave instruction serves the 64 permutations of its lanes)
        out["clocks_per_valu_instruction_and_simd"] = simd_clocks / (permutations_per_gpu_step / 64.0 * per)     # ~4.0 = every issue slot taken
    return out


def fmt_pow2(n):
    return f"2^{n.bit_length() - 1}" if n and n & (n - 1) == 0 else str(n)


def run_verification(env):
    """One untimed pass of the same step functions' engine over a fresh copy of the seeded inputs; a sample of the
    results against the C restatement (oracle/), per rank.  At N > 1 the permutation workloads check the GATHERED buffer:
    rank r's shard must sit at offset r's span, whoever did the gather."""
    import torch
    from sponge_amd import mgpu, synth
    import sponge_amd as S
    args, world, rank, dev = env["args"], env["world"], env["rank"], env["dev"]
    ctx, group, stream = env["ctx"], env["group"], env["stream"]
    field, field_name, t = env["field"], env["field_name"], env["t"]
    rate, alpha, rf, rp, seed = env["rate"], env["alpha"], env["rf"], env["rp"], env["seed"]
    n, n_total, start = env["n"], env["n_total"], env["start"]
    cr = oracle_engine(field_name, rate, alpha, rf, rp)
    res = {"ok": True, "engine": "oracle/poseidon_ref.c (C restatement)", "what": None}

    def to_np(x):
        return x.cpu().numpy().view(np.uint64)

    if env["duplex"]:
        # two steps from fresh sponges (the second one runs the steady cycle the timed region runs), a sample sponge by sponge
        in_len, out_len = env["in_len"], env["out_len"]
        host, d_in = env["fresh_inputs"]()
        st = torch.zeros((n, t, 4), dtype=torch.int64, device=dev)
        tag = torch.zeros(n, dtype=torch.int32, device=dev)
        idx_t = torch.zeros(n, dtype=torch.int32, device=dev)
        d_out = torch.zeros((n, out_len, 4), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for _ in range(2):
            ctx.sponge_absorb_batch_dev(st.data_ptr(), tag.data_ptr(), idx_t.data_ptr(), d_in.data_ptr(), in_len, n, stream.cuda_stream)
            ctx.sponge_squeeze_batch_dev(st.data_ptr(), tag.data_ptr(), idx_t.data_ptr(), d_out.data_ptr(), out_len, n, stream.cuda_stream)
        torch.cuda.synchronize()
        idx = sample_indices(n, 512)
        got_st, got_out = to_np(st).reshape(n, t, 4), to_np(d_out).reshape(n, out_len, 4)
        rows = host.reshape(n, in_len, 4)
        ok = True
        for j in idx:
            s_j, m_j, i_j = np.zeros((t, 4), dtype=np.uint64), 0, 0
            for _ in range(2):
                s_j, m_j, i_j = cr.sponge_absorb(s_j, m_j, i_j, rows[j])
                s_j, m_j, i_j, o_j = cr.sponge_squeeze(s_j, m_j, i_j, out_len)
            ok &= bool(np.array_equal(got_st[j], s_j) and np.array_equal(got_out[j], o_j) and int(tag[j]) == m_j and int(idx_t[j]) == i_j)
        res["ok"] = ok
        res["what"] = f"{len(idx)} of this rank's {n} sponges after two absorb + squeeze steps from fresh sponges: state, output and mode words"
    elif env["hashing"]:
        in_len, out_len = env["in_len"], env["out_len"]
        host, d_in = env["fresh_inputs"]()
        d_out = torch.zeros((n, out_len, 4), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()     # the fills above ran on torch's stream; the library's stream is non-blocking
        ctx.hash_batch_dev(d_in.data_ptr(), in_len, d_out.data_ptr(), out_len, n, stream.cuda_stream)
        torch.cuda.synchronize()
        idx = sample_indices(n, 2048)
        want = cr.hash_batch(np.ascontiguousarray(host.reshape(n, in_len, 4)[idx]), in_len, out_len, threads=0)
        res["ok"] = bool(np.array_equal(to_np(d_out).reshape(n, out_len, 4)[idx], want))
        res["what"] = f"{len(idx)} of this rank's {n} rows of one fresh launch"
    elif env["merkle"]:
        host, d_leaves = env["fresh_inputs"]()
        nodes = torch.zeros((2 * n - 1, 4), dtype=torch.int64, device=dev)
        nodes[:n] = d_leaves.reshape(n, 4)
        top = torch.zeros((max(2 * world - 1, 1), 4), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()     # leaves copied and buffers zeroed (torch's stream) before the library's stream reads them
        if group is not None:
            group.merkle_2to1_dev([nodes.data_ptr()], [top.data_ptr()], n_total)
        else:
            ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)
        torch.cuda.synchronize()
        got = to_np(nodes)
        ok = True
        # level 1 on a sample of pairs
        if n >= 2:
            idx = sample_indices(n // 2, 2048)
            pairs = np.ascontiguousarray(host.reshape(n // 2, 2, 4)[idx])
            ok &= bool(np.array_equal(got[n + idx], cr.hash_batch(pairs, 2, 1, threads=0).reshape(-1, 4)))
        # the top of this rank's subtree, rebuilt by the checker from the GPU's own level of <= 1024 nodes
        w = min(n, 1024)
        first = 2 * n - 2 * w                                 # offset of the level that has w nodes
        sub = cr.merkle(np.ascontiguousarray(got[first:first + w]), threads=0)
        ok &= bool(np.array_equal(sub[w:], got[first + w:]))
        if world > 1:
            gtop = to_np(top)
            ok &= bool(np.array_equal(gtop[rank], got[-1]))                       # my root at my rank's slot
            ok &= bool(np.array_equal(cr.merkle(np.ascontiguousarray(gtop[:world]), threads=0), gtop))   # the top log2 N levels
        res["ok"] = ok
        res["what"] = (f"a fresh tree: level 1 on a sample, the top {w.bit_length() - 1} levels of this rank's subtree"
                       + (f", the {world} gathered roots and the levels above them" if world > 1 else ""))
    else:
        host, fresh = env["fresh_inputs"]()
        fresh = fresh.reshape(n, t, 4)
        gathered = env.get("gathered")
        if world > 1 and gathered is not None:
            gathered.zero_()
        torch.cuda.synchronize()     # upload and fill (torch's stream) complete before the library's non-blocking stream starts
        to_root, overlapped = args.gather in ("root", "overlap-root"), args.gather in ("overlap", "overlap-root")
        dst = [gathered.data_ptr() if gathered is not None else 0]
        if group is not None and overlapped:
            group.permute_gather_dev([fresh.data_ptr()], dst, n_total, 0 if to_root else -1, args.gather_chunks)
        elif group is not None:
            group.permute_shards_dev([fresh.data_ptr()], n_total)
        else:
            ctx.permute_batch_dev(fresh.data_ptr(), n, stream.cuda_stream)
        if world > 1 and args.gather != "none" and not overlapped:      # (every rank takes part, whoever receives)
            if to_root:
                group.gather_dev([fresh.data_ptr()], dst, n_total, t, 0)
            else:
                group.all_gather_dev([fresh.data_ptr()], dst, n_total, t)
        if world > 1 and gathered is not None:
            torch.cuda.synchronize()
            ok, checked = True, 0
            for r in range(world):               # every rank checks every shard of ITS copy of the gathered result
                s_r, c_r = mgpu.shard_bounds(n_total, world, r)
                idx = s_r + sample_indices(c_r, 256)
                inp = np.stack([synth.random_elements(field, t, seed, offset=int(i) * t) for i in idx])
                ok &= bool(np.array_equal(to_np(gathered[torch.from_numpy(idx).to(dev)]), cr.permute_batch(inp, threads=0)))
                checked += len(idx)
            res["ok"] = ok
            res["what"] = f"gathered buffer of one fresh pass: {checked} states, a sample of every rank's span at its offset"
        else:
            torch.cuda.synchronize()
            idx = sample_indices(n, 4096)
            want = cr.permute_batch(np.ascontiguousarray(host.reshape(n, t, 4)[idx]), threads=0)
            res["ok"] = bool(np.array_equal(to_np(fresh)[idx], want))
            res["what"] = f"{len(idx)} of this rank's {n} states of one fresh launch"
    return res


def evidence_stale(recs):
    """The committed counter figures belong to the device code they were taken on: tools/source_hash.py's hash of sponge_amd/csrc is stored
    with them.  A tree whose kernels have changed since must not quote them (tests/test_evidence_fresh.py fails on such a tree)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import kernel_source_hash
        have, want = recs.get("_kernel_source_hash"), kernel_source_hash()
        return None if have == want else "stale: taken on kernel sources %s, this tree is %s - re-run the counter passes (tools/gpu_r06.sh slots pmc)" % (have, want)
    except Exception as e:   # (a missing helper is not a reason to quote unverifiable figures)
        return "stale: cannot hash the kernel sources (%s)" % e


def load_traffic(workload, per_gpu_units):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of this command (profiles/hbm_traffic.json); it
    cannot be collected inside a timed run (counter collection serialises the kernels).  Only quoted when the profile was
    taken at this run's size; otherwise the line says that nothing was measured for this workload."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        recs = json.load(open(path))
        stale = evidence_stale(recs)
        if stale:
            return None, stale
        rec = next((r for k, r in recs.items() if isinstance(r, dict) and k.split("_")[0] == workload and abs(r.get("units_per_launch", per_gpu_units) - per_gpu_units) <= 1), None)
        if rec is None:
            return None, "not measured for this workload at this size"
        return rec["bytes_per_launch"], ("profiles/hbm_traffic.json: a committed constant, not a measurement of this run (separate rocprofv3 --pmc "
                                         "FETCH_SIZE / WRITE_SIZE passes of this command, %s)" % rec.get("taken", "round 3"))
    except Exception:
        return None, "not measured for this workload"


def load_valu_issue(workload, per_gpu_units, kernel_s, compute_units, mads, info, slot):
    """The binding resource, one level below `int_valu`: a SIMD of this part takes ONE VALU instruction per ~4 shader clocks
    from any stream that holds multiplies, whatever the instruction is and whichever wave it comes from
    (tools/issue_model_microbench.hip -> profiles/r03/f_issue_model_microbench.txt), so a kernel's floor is its VALU
    instruction count, not its multiply count.  The slot is measured IN THIS RUN on this device at the kernel's occupancy
    (pmx_diag_issue_slot: three calibration streams, the fastest is the floor); the instruction count per permutation is a
    property of the build and comes from the committed SQ_INSTS_VALU pass of this command (profiles/valu_instructions.json)."""
    waves = max(1, min(8, info.waves_per_simd))
    r = slot.get(waves)
    out = {"bound": "VALU issue slots (one instruction per SIMD and ~4 clocks in a stream with multiplies)", "waves_per_simd": waves}
    if r is not None:
        out["calibration_ns_per_instruction_and_simd"] = {"12 mad + 4 simple (the kernels' mix)": r.ns_12mad_4simple, "4 mad + 12 simple": r.ns_4mad_12simple,
                                                          "16 mad": r.ns_16mad}
        out["floor_ns_per_instruction_and_simd"] = r.ns_floor
        out["floor_source"] = "pmx_diag_issue_slot on this device before the warm-up, %d launches, exactly %d waves resident per SIMD; floor = the fastest stream" % (r.launches, waves)
    try:
        recs = json.load(open(os.path.join(ROOT, "profiles", "valu_instructions.json")))
        stale = evidence_stale(recs)
        if stale:
            out.update({"valu_instructions_per_permutation": None, "frac": None, "count_source": stale})
            return out
        w = recs[workload]
        # (the count per permutation is a property of the kernel, not of the batch size: quoted whenever the engine is the one the pass was taken on)
        if not compute_units or w.get("engine", info.engine.decode()) != info.engine.decode():
            raise KeyError(workload)
        per_unit = w.get("valu_instructions_per_permutation", w["valu_instructions_per_wave"])   # what a lane executes per permutation
        ns = kernel_s * 1e9 * compute_units * 4 / (per_unit * per_gpu_units / 64.0)
        out.update({"valu_instructions_per_permutation": per_unit, "multiply_share": mads / per_unit, "ns_per_instruction_and_simd": ns,
                    "count_source": "profiles/valu_instructions.json (rocprofv3 --pmc SQ_INSTS_VALU pass of this command, " + w.get("source", "") + ")"})
        if r is not None:
            out["frac"] = r.ns_floor / ns
            out["ratio_to_own_mix_stream"] = r.ns_12mad_4simple / ns    # a calibration, not a bound: may exceed 1 by the noise of two timings
    except Exception:
        out["valu_instructions_per_permutation"] = None
        out["frac"] = None
        if workload in ("c5", "d3", "d9"):
            out["count_source"] = ("not quoted: a step of this workload is several kernels (one per tree level / pass 0 and the listed passes of an absorb and of a "
                                   "squeeze), so there is no single per-permutation count; their SQ counters per kernel: profiles/r05/z_valu_driver_and_tree_kernels.txt")
        else:
            out["count_source"] = "no SQ_INSTS_VALU pass committed for this workload / engine"
    return out


if __name__ == "__main__":
    main()
