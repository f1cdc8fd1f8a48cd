"""One rank of a multi-process device group (a child process of tests/test_gpu_mgpu.py; never collected by pytest).

    python tests/mgpu_rank_worker.py RANK WORLD DEVICE UID_FILE N_TOTAL LOG2_LEAVES_PER_RANK OUT_JSON

Everything goes through the C ABI as a Rust / C++ caller would use it - pmx_mgpu_unique_id / pmx_mgpu_create_rank
(ncclCommInitRank), the library's own device-memory helpers, pmx_mgpu_permute_shards_dev, pmx_mgpu_all_gather_dev,
pmx_mgpu_gather_dev, pmx_mgpu_permute_gather_dev, pmx_mgpu_merkle_2to1_dev - with no torch in the process.  Rank 0 makes the communicator id and hands it to the other
ranks through UID_FILE.  Each rank then checks ITS copy of the gathered buffer (every rank's span, in full) and the
sharded Merkle root against the C restatement and writes a verdict to OUT_JSON."""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, device = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    uid_file, n_total, log2_leaves, out_json = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    res = {"rank": rank, "ok": False, "error": None}
    try:
        import sponge_amd as S
        from sponge_amd import _lib, mgpu, synth
        if os.environ.get("PMX_RCCL_LIBRARY"):      # ranks sharing one GPU behind the named stand-in: only the test-hook build reads that variable
            _lib.use_test_library()
        from gpu_helpers import c_oracle, product_config
        lib = _lib.lib()
        name = "bls_t3_a5_8_31"
        cfg = product_config(name)
        t = 3
        if rank == 0:
            uid = mgpu.unique_id()
            with open(uid_file + ".tmp", "wb") as f:
                f.write(uid)
            os.rename(uid_file + ".tmp", uid_file)          # atomic: the other ranks never see half an id
        else:
            t0 = time.time()
            while not os.path.exists(uid_file):
                if time.time() - t0 > 300:
                    raise RuntimeError("rank 0 never published the communicator id")
                time.sleep(0.05)
            uid = open(uid_file, "rb").read()
        g = mgpu.DeviceGroup.one_rank(cfg, device, rank, world, uid)
        info = g.info()
        res["info"] = {k: info[k] for k in ("world", "n_local", "first_rank", "comm_ranks", "comm_first_rank", "rccl_version_str")}
        assert info["comm_ranks"] == world and info["comm_first_rank"] == rank and info["n_local"] == 1

        def dalloc(nbytes):
            p = ctypes.c_void_p()
            _lib.check(lib.pmx_device_alloc(device, ctypes.byref(p), nbytes))
            return p

        def upload(dst, arr):
            _lib.check(lib.pmx_device_upload(device, dst, ctypes.c_void_p(arr.ctypes.data), arr.nbytes, ctypes.c_void_p(g.stream(0))))

        def download(arr, src):
            _lib.check(lib.pmx_device_download(device, ctypes.c_void_p(arr.ctypes.data), src, arr.nbytes, ctypes.c_void_p(g.stream(0))))
            _lib.check(lib.pmx_stream_synchronize(device, ctypes.c_void_p(g.stream(0))))

        # ---- sharded permutation + the RCCL gather (equal or ragged by n_total) ----------------------------------------
        start, count = mgpu.shard_bounds(n_total, world, rank)
        seed = 0x5EED0050
        mine = synth.random_elements(cfg.field, max(count, 1) * t, seed, offset=start * t).reshape(-1, t, 4)[:count]
        d_shard, d_all = dalloc(max(count, 1) * t * 32), dalloc(n_total * t * 32)
        if count:
            upload(d_shard, np.ascontiguousarray(mine))
        upload(d_all, np.zeros((n_total, t, 4), dtype=np.uint64))
        g.permute_shards_dev([d_shard.value], n_total)
        g.all_gather_dev([d_shard.value], [d_all.value], n_total, t)
        got = np.zeros((n_total, t, 4), dtype=np.uint64)
        download(got, d_all)
        whole = synth.random_elements(cfg.field, n_total * t, seed).reshape(n_total, t, 4)
        want = c_oracle(name).permute_batch(whole, threads=0)
        bad = []
        for r in range(world):              # every rank's span of MY gathered copy, in full
            s_r, c_r = mgpu.shard_bounds(n_total, world, r)
            if not np.array_equal(got[s_r:s_r + c_r], want[s_r:s_r + c_r]):
                bad.append(r)
        res["gather_bad_spans"] = bad
        # ---- the same result to ONE rank (grouped ncclSend / ncclRecv), then the last step with its gather piece by piece ------------
        p2p_bad = []
        root = world - 1
        if count:
            upload(d_shard, np.ascontiguousarray(mine))
        upload(d_all, np.zeros((n_total, t, 4), dtype=np.uint64))
        g.permute_shards_dev([d_shard.value], n_total)
        g.gather_dev([d_shard.value], [d_all.value if rank == root else 0], n_total, t, root)
        download(got, d_all)
        if not np.array_equal(got, want if rank == root else np.zeros_like(want)):      # only the root's buffer is written
            p2p_bad.append("gather_dev")
        for gather_root, chunks in ((-1, 4), (0, 3)):
            if count:
                upload(d_shard, np.ascontiguousarray(mine))
            upload(d_all, np.zeros((n_total, t, 4), dtype=np.uint64))
            receives = gather_root < 0 or gather_root == rank
            g.permute_gather_dev([d_shard.value], [d_all.value if receives else 0], n_total, gather_root, chunks)
            download(got, d_all)
            if not np.array_equal(got, want if receives else np.zeros_like(want)):
                p2p_bad.append(f"permute_gather_dev(root {gather_root}, {chunks} pieces)")
        res["p2p_bad"] = p2p_bad
        # ---- sharded Merkle tree: subtree per rank, all-gather of the roots, top levels on every rank --------------------
        tree_ok = None
        if world & (world - 1) == 0:
            m = 1 << log2_leaves
            leaves_all = synth.random_elements(cfg.field, world * m, seed + 1)
            d_nodes, d_top = dalloc((2 * m - 1) * 32), dalloc(max(2 * world - 1, 1) * 32)
            upload(d_nodes, np.ascontiguousarray(leaves_all[rank * m:(rank + 1) * m]))
            g.merkle_2to1_dev([d_nodes.value], [d_top.value], world * m)
            top = np.zeros((max(2 * world - 1, 1), 4), dtype=np.uint64)
            download(top, d_top)
            want_nodes = c_oracle(name).merkle(leaves_all, threads=0)
            tree_ok = bool(np.array_equal(top[-1], want_nodes[-1]))
            if world > 1:      # the gathered subtree roots are the level of the whole tree that has `world` nodes
                tree_ok = tree_ok and bool(np.array_equal(top[:world], want_nodes[2 * world * m - 2 * world: 2 * world * m - world]))
            for p in (d_nodes, d_top):
                lib.pmx_device_free(device, p)
        res["tree_ok"] = tree_ok
        g.synchronize()
        for p in (d_shard, d_all):
            lib.pmx_device_free(device, p)
        g.close()
        res["ok"] = (not bad) and (not p2p_bad) and tree_ok is not False
    except Exception as e:       # noqa: BLE001
        import traceback
        res["error"] = repr(e) + "\n" + traceback.format_exc()[-1500:]
    with open(out_json, "w") as f:
        json.dump(res, f)
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
