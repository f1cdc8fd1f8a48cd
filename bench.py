#!/usr/bin/env python3
"""Benchmark of the batched Poseidon permutation on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (pmx_permute_batch_dev: PoseidonSponge::permute on every state,
reference src/poseidon/mod.rs:95-118) over one device-resident batch of synthetic random states.
Workload at N=1: BASELINE.json configs[1] -- 2^20 independent states, BLS12-381 Fr, t=3, alpha=5, 8+31
rounds.  For N>1 (launched by torch.distributed.run, one rank per GPU) every rank permutes its own
2^20-state shard (weak scaling, NO collective on the data path); after the K steps the result shards are
all-gathered once over RCCL inside the timed region (the "final gather"; --gather overlap|serial gathers after
every step instead).

Rank 0 prints ONE JSON line.  `value` = permutations per second over all ranks, inputs already in HBM.
`roofline` prices the permutation kernel against HBM (algorithmic 2*t*32 bytes per permutation);
`cpu_baseline` times the C restatement of the reference algorithm (oracle/, kind "port") on the host
cores for a bounded sample of the same workload (rank 0, N=1 only).
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

VALU_MAD_PEAK = 3.19e13       # v_mad_u64_u32 lane-instructions/s, measured (profiles/r01/valu_microbench.txt)

WORKLOADS = {
    # name: (field, rate, alpha, RF, RP, log2 units per GPU, seed, description)
    "c2": ("bls12_381_fr", 2, 5, 8, 31, 20, 0x5EED0002, "bls12_381_fr t=3 alpha=5 RF=8 RP=31, 2^20 states/GPU"),
    "c3": ("bn254_fr", 8, 5, 8, 57, 18, 0x5EED0003, "bn254_fr t=9 alpha=5 RF=8 RP=57, 2^18 states/GPU"),
    # 2-to-1 Merkle compression: every rank reduces its own 2^21-leaf subtree level by level (2^21 - 1
    # permutations), the subtree roots are all-gathered and the top log2(N) levels finished on every rank
    "c5": ("bls12_381_fr", 2, 5, 8, 31, 21, 0x5EED0005, "bls12_381_fr t=3 alpha=5 2-to-1 Merkle tree, 2^21 leaves/GPU"),
    # the absorb/squeeze batch driver (pmx_hash_batch_dev): per row new; absorb(L); squeeze_native(1)
    "h3": ("bls12_381_fr", 2, 5, 8, 31, 20, 0x5EED0006, "bls12_381_fr t=3 alpha=5 hash of 4 elements -> 1 (2 permutations/row), 2^20 rows/GPU"),
    "h9": ("bn254_fr", 8, 5, 8, 57, 18, 0x5EED0007, "bn254_fr t=9 alpha=5 hash of 8 elements -> 1 (1 permutation/row), 2^18 rows/GPU"),
}
HASH_SHAPES = {"h3": (4, 1, 2), "h9": (8, 1, 1)}     # workload -> (in_len, out_len, permutations per row)


def mads_per_permutation(t, alpha, rf, rp, optimised, row_tables=False, lane_tables=False):
    """v_mad_u64_u32 count of one permutation as implemented (pmx_field.hpp): product 81, square 45, reduction 81 limb
    products; one reduction per S-box step and per matrix row.  With shifted tables (tab_dot) a row of N constants
    costs 81 N + 18 instead of 81 N + 81 (`row_tables`) and an identity-lane update 81 + 18 + 9 instead of 162 + 9
    (`lane_tables`)."""
    sqr, mul = 45 + 81, 81 + 81
    chain = {5: 2 * sqr + mul, 17: 4 * sqr + mul}.get(alpha)
    if chain is None:
        bits = bin(alpha)[3:]
        chain = len(bits) * sqr + bits.count("1") * mul
    dot = 81 * t + (18 if row_tables else 81)
    lane = (81 + 18 + 9) if lane_tables else (mul + 9)     # + 9 multiply-by-one injections of the addend
    full = t * chain + t * dot
    if optimised:
        partial = (rp - 1) * (chain + dot + (t - 1) * lane) + (chain + t * dot)
    else:
        partial = rp * (chain + t * dot)
    return rf * full + partial


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--states-per-gpu-log2", type=int, default=None)
    ap.add_argument("--gather", default="final", choices=["final", "overlap", "serial", "none"],
                    help="N>1: 'final' = one RCCL all-gather of the result shards after the K steps (inside the timed "
                         "region); 'overlap'/'serial' = an all-gather after EVERY step (double-buffered / blocking)")
    ap.add_argument("--spinup-seconds", type=float, default=0.25, help="untimed device spin-up before the W warmup steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def cpu_baseline(field_name, rate, alpha, rf, rp, seed, target_seconds):
    """C restatement of the reference permutation (oracle/poseidon_ref.c) on all host cores."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    import sponge_amd as S
    from sponge_amd import synth

    p, bits = {"bls12_381_fr": (O.BLS12_381_FR, 255), "bn254_fr": (O.BN254_FR, 254)}[field_name]
    ocfg = O.make_config(p, bits, rate, alpha, rf, rp)
    cr = cref.CRef(ocfg)
    t = rate + 1
    threads = cref.max_threads()      # affinity mask capped by the cgroup CPU quota
    field = S.FIELDS[field_name]
    probe = synth.random_elements(field, 4096 * t, seed).reshape(4096, t, 4)
    t0 = time.perf_counter()
    cr.permute_batch(probe, threads=threads)
    dt = max(time.perf_counter() - t0, 1e-6)
    n = int(min(1 << 22, max(4096, 4096 * target_seconds / dt)))
    n = 1 << (n.bit_length() - 1)
    batch = synth.random_elements(field, n * t, seed).reshape(n, t, 4)
    t0 = time.perf_counter()
    cr.permute_batch(batch, threads=threads)
    dt = time.perf_counter() - t0
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": n / dt, "unit": "permutations/s", "cores": threads, "kind": "port",
            "sample": f"first {n} states of the same seeded batch, C restatement of mod.rs:63-118 "
                      f"(dense MDS, square-and-multiply pow), OpenMP x{threads}, {dt:.2f} s, CPU: {model}"}


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Poseidon path has no CPU fallback")

    import torch.distributed as dist
    import sponge_amd as S
    from sponge_amd import synth

    # PMX_BENCH_BACKEND=gloo is a single-GPU rehearsal of the N > 1 code path (tools/gpu_n2_rehearsal.sh): the ranks
    # share the visible GPUs round-robin and the gathers are staged through the host.  Its numbers mean nothing.
    backend = os.environ.get("PMX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    field_name, rate, alpha, rf, rp, log2n, seed, desc = WORKLOADS[args.workload]
    if args.states_per_gpu_log2 is not None:
        log2n = args.states_per_gpu_log2
    n = 1 << log2n
    t = rate + 1
    field = S.FIELDS[field_name]
    cfg = S.poseidon_config_from_lfsr(field, rate, alpha, rf, rp)
    ctx = cfg.context(local_rank)

    from sponge_amd import distributed as D

    stream = torch.cuda.current_stream()
    merkle = args.workload == "c5"
    hashing = args.workload in HASH_SHAPES
    if hashing:
        in_len, out_len, perms_per_row = HASH_SHAPES[args.workload]
        host = synth.random_elements(field, n * in_len, seed, offset=rank * n * in_len)
        d_in = torch.from_numpy(host.view(np.int64).copy()).to(dev)
        d_out = torch.zeros((n, out_len, 4), dtype=torch.int64, device=dev)
        units_per_step = float(world) * n * perms_per_row

        def step(i):
            ctx.hash_batch_dev(d_in.data_ptr(), in_len, d_out.data_ptr(), out_len, n, stream.cuda_stream)

        def drain():
            pass

        def final_gather():
            pass
    elif merkle:
        # leaves of this rank's subtree, resident in the first n rows of the node array [2n-1][4]
        host = synth.random_elements(field, n, seed, offset=rank * n)
        nodes = torch.zeros((2 * n - 1, 4), dtype=torch.int64, device=dev)
        nodes[:n] = torch.from_numpy(host.view(np.int64).copy()).to(dev)
        top = torch.zeros((2 * world - 1, 4), dtype=torch.int64, device=dev)
        units_per_step = float(world) * (n - 1) + (world - 1)          # permutations per step, all ranks

        def step(i):
            ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)
            if world > 1:
                top[:world] = D.all_gather_equal(nodes[2 * n - 2:2 * n - 1])   # 32-byte subtree roots
                ctx.merkle_2to1_dev(top.data_ptr(), world, stream.cuda_stream)

        def drain():
            pass

        def final_gather():
            pass
    else:
        # this rank's shard of the global seeded batch [world*n][t][4]
        host = synth.random_elements(field, n * t, seed, offset=rank * n * t)
        per_step = world > 1 and args.gather in ("overlap", "serial")
        n_buf = 2 if (world > 1 and args.gather == "overlap") else 1
        bufs = [torch.from_numpy(host.view(np.int64).copy()).to(dev).reshape(n, t, 4) for _ in range(n_buf)]
        gathered = [torch.empty((world * n, t, 4), dtype=torch.int64, device=dev) for _ in range(n_buf)] \
            if (world > 1 and args.gather != "none") else None
        pending = [None] * n_buf
        units_per_step = float(world) * n

        def step(i):
            b = i % n_buf
            if pending[b] is not None:           # the gather that still reads this buffer
                pending[b].wait()
                pending[b] = None
            ctx.permute_batch_dev(bufs[b].data_ptr(), n, stream.cuda_stream)   # no collective on the data path
            if per_step:
                _, work = D.all_gather_equal(bufs[b], out=gathered[b], async_op=True)
                if args.gather == "serial":
                    work.wait()
                else:
                    pending[b] = work

        def drain():
            for b in range(n_buf):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None

        def final_gather():                      # the job's epilogue: every rank ends with the whole result
            if world > 1 and args.gather == "final":
                D.all_gather_equal(bufs[0], out=gathered[0])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Device spin-up (untimed, not counted as steps): the shader clock ramps over the first ~25 ms of activity
    # (per-launch time falls from 2.5 ms to 2.07 ms across the first dozen launches, profiles/r01), so the W warmup
    # steps alone would leave short runs measuring the ramp instead of the kernel.
    t_spin = time.perf_counter()
    i_spin = 0
    # a step that contains a collective must run the same number of times on every rank: fixed count instead of a clock
    collective_in_step = world > 1 and (merkle or args.gather in ("overlap", "serial"))
    while (i_spin < 64) if collective_in_step else (time.perf_counter() - t_spin < args.spinup_seconds):
        step(i_spin)
        i_spin += 1
        if i_spin % 4 == 0:
            drain()
            torch.cuda.synchronize()
    drain()
    for i in range(args.warmup):
        step(i)
    drain()
    final_gather()      # also warms RCCL's lazily built rings up, outside the timed region
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for i in range(args.steps):
        step(i)
    drain()
    final_gather()
    ev1.record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)          # HIP events on the launch stream

    times = torch.tensor([elapsed, dev_ms / 1e3], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    elapsed, dev_s = float(times[0]), float(times[1])

    if rank == 0:
        value = units_per_step * args.steps / elapsed
        # dominant kernel: permute_kernel (hash_kernel in the Merkle mode); its average launch duration from the
        # HIP events on this rank's launch stream (at N=1 the timed region holds nothing but the back-to-back launches)
        kernel_s = dev_s / args.steps
        per_gpu_units = units_per_step / world
        bytes_per_unit = 96 if merkle else 2 * t * 32       # 2*t*32 B per permutation; 64 in + 32 out per 2-to-1
        if hashing:
            bytes_per_unit = (in_len + out_len) * 32 / perms_per_row
        algo_bytes = bytes_per_unit * per_gpu_units
        achieved = algo_bytes / kernel_s / 1e9
        # engines as dispatched (pmx_device.hip): t = 3..9 run the optimised schedule, with every matrix as shifted tables up
        # to t = 5 and the identity lanes only above
        mads = mads_per_permutation(t, alpha, rf, rp, optimised=3 <= t <= 9, row_tables=3 <= t <= 5, lane_tables=3 <= t <= 9) \
            + (3 if merkle else 2 * t) * 162   # + ABI conversions
        mad_rate = mads * per_gpu_units / kernel_s
        out = {
            "metric": "Poseidon permutations/sec (%s, t=%d)" % ({"bls12_381_fr": "BLS12-381 Fr", "bn254_fr": "BN254 Fr"}[field_name], t),
            "value": value, "unit": "permutations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": desc, "arithmetic": "255-bit modular integers as 9 x 29-bit limbs in u32, Montgomery form",
                       "units_per_gpu": n, "permutations_per_step": units_per_step,
                       "gather": (args.gather if (world > 1 and not merkle) else ("roots" if merkle and world > 1 else "n/a")),
                       "sharding": f"contiguous x{world}",
                       **({"rehearsal_backend": backend} if backend != "nccl" else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(args.workload),
                         "kernel": "pmx::compress_kernel / compress_coop_kernel (per tree level)" if merkle else
                                   ("pmx::hash_kernel" if hashing else "pmx::permute_kernel"),
                         "kernel_ms": 1e3 * kernel_s, "algorithmic_bytes_per_launch": algo_bytes,
                         "note": "integer-VALU bound, not HBM bound (DESIGN.md): see int_valu"},
            "int_valu": {"bound": "v_mad_u64_u32 issue", "achieved": mad_rate, "peak": VALU_MAD_PEAK,
                         "unit": "lane-instr/s", "frac": mad_rate / VALU_MAD_PEAK, "mads_per_permutation": mads},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(field_name, rate, alpha, rf, rp, seed, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def load_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/hbm_traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        return json.load(open(path))[workload]["bytes_per_launch"]
    except Exception:
        return None


if __name__ == "__main__":
    main()
