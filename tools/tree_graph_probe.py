"""Does a hipGraph shorten the 21 dependent launches of a 2^21-leaf tree?  pmx_merkle_2to1_dev only enqueues kernels on the caller's stream
(no allocation, no synchronisation), so a caller may capture it; this probe times 20 trees launched level by level on a stream against 20
replays of one captured tree, and the same for the 15 narrow levels alone (a 2^15-leaf tree).  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sponge_amd as S  # noqa: E402
from sponge_amd import synth  # noqa: E402

field = S.FIELDS["bls12_381_fr"]
cfg = S.poseidon_config_from_lfsr(field, 2, 5, 8, 31)
ctx = cfg.context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream()
for log2 in (21, 15, 10):
    n = 1 << log2
    host = synth.random_elements(field, n, 77)
    nodes = torch.zeros((2 * n - 1, 4), dtype=torch.int64, device=dev)
    nodes[:n] = torch.from_numpy(host.view(np.int64).copy()).to(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(30 if log2 == 21 else 200):
            ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)
        stream.synchronize()
        want = nodes[-1].clone()
        reps = 20 if log2 == 21 else 100

        def timed(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record(stream)
                for _ in range(reps):
                    fn()
                e1.record(stream)
                stream.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps)
            return best
        t_stream = timed(lambda: ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream))
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=stream):
                ctx.merkle_2to1_dev(nodes.data_ptr(), n, stream.cuda_stream)
            nodes[-1].zero_()
            t_graph = timed(g.replay)
            ok = bool(torch.equal(nodes[-1], want))
            print("2^%d leaves: %d launches on the stream %.4f ms per tree, one captured graph replayed %.4f ms (%+.1f %%), root equal: %s"
                  % (log2, log2, t_stream, t_graph, 100 * (t_stream / t_graph - 1), ok), flush=True)
        except Exception as e:   # noqa: BLE001
            print("2^%d leaves: stream %.4f ms; capture failed: %r" % (log2, t_stream, e), flush=True)
