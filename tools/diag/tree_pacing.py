#!/usr/bin/env python3
"""How the time of one 2^21-leaf tree (21 launches) depends on the number of trees enqueued back to back without a host sync:
`bench.py --workload c5 --total-log2 21 --steps 30` read 3.9997 ms on two boxes for two builds whose kernels differ by 1.5 %."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sponge_amd as S
from sponge_amd import synth
cfg = S.poseidon_config_from_lfsr(S.BLS12_381_FR, 2, 5, 8, 31)
ctx = cfg.context()
n = 1 << 21
dev = torch.device("cuda:0")
nodes = torch.zeros((2 * n - 1, 4), dtype=torch.int64, device=dev)
nodes[:n] = torch.from_numpy(synth.random_elements(S.BLS12_381_FR, n, seed=1).view("int64")).to(dev)
stream = torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(3):
    ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)
torch.cuda.synchronize()
for steps in (1, 5, 10, 15, 20, 24, 25, 26, 30, 40, 60):
    for rep in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            e0.record(stream)
            for _ in range(steps):
                ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)
            e1.record(stream)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("steps %3d  gpu %.4f ms/tree   host enqueue %.4f ms/tree   wall %.4f ms/tree" % (steps, e0.elapsed_time(e1) / steps, (t1 - t0) * 1e3 / steps, (t2 - t0) * 1e3 / steps), flush=True)
