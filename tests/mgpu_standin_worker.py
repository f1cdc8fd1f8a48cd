"""Every world > 1 branch of the device-group code (sponge_amd/csrc/pmx_mgpu.cpp) on ONE GPU, through the real C ABI,
behind the stand-in collective library tests/fake_rccl (a child process of tests/test_gpu_mgpu_standin.py and of
tools/mgpu_coverage.sh; never collected by pytest).

    LD_LIBRARY_PATH=tests/fake_rccl:$LD_LIBRARY_PATH \
        python tests/mgpu_standin_worker.py WORLD OUT_JSON [LIBRARY]

The product binds RCCL with dlopen("librccl.so.1"): with tests/fake_rccl first on LD_LIBRARY_PATH it gets the stand-in
(asserted below through fake_rccl_marker - a run that silently bound the real RCCL fails).  The test hook
pmx_mgpu_test_shared_device lets pmx_mgpu_create take device 0 in all WORLD slots: it exists only in
libposeidon_mi355x_test.so (the shipped objects + pmx_mgpu.cpp compiled -DPMX_TEST_HOOKS), which is what this process binds.
No torch in this process (torch would map its own librccl first).  LIBRARY (optional) = another test-hook build of the
library, e.g. the one with gcov counters in pmx_mgpu.o.

Each scenario appends {"name", "ok", "detail"} to OUT_JSON["scenarios"]; the reference has no counterpart for any of
this (src/poseidon/mod.rs:62-183: states are independent, which is the whole contract the sharding relies on) - the
expected values are the C restatement's over the WHOLE batch / tree, whatever the sharding."""
import ctypes
import json
import os
import sys
import threading
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAME = "bls_t3_a5_8_31"
T = 3


def main():
    world, out_json = int(sys.argv[1]), sys.argv[2]
    res = {"world": world, "scenarios": [], "ok": False, "error": None}

    def record(name, ok, detail=""):
        res["scenarios"].append({"name": name, "ok": bool(ok), "detail": str(detail)[-1500:]})
        print(("ok      " if ok else "FAILED  ") + name, flush=True)      # (a run that hangs shows how far it came)

    try:
        from sponge_amd import _lib
        if len(sys.argv) > 3:
            _lib.use_library(sys.argv[3], test_hooks=True)
        else:
            _lib.use_test_library()
        import sponge_amd as S
        from sponge_amd import mgpu, synth
        from gpu_helpers import c_oracle, product_config
        lib = _lib.lib()
        res["library"] = _lib.library_path()
        assert lib.pmx_test_hooks_enabled() == 1
        cfg = product_config(NAME)
        cr = c_oracle(NAME)
        # the copy the product binds: the one PMX_RCCL_LIBRARY names, else the same SONAME on the same search path
        fake = ctypes.CDLL(os.environ.get("PMX_RCCL_LIBRARY") or "librccl.so.1")
        assert fake.fake_rccl_marker() == 0x5EED, "the real RCCL is on the path, not tests/fake_rccl"
        fake.fake_rccl_stats.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
        fake.fake_rccl_stats.restype = None
        fake.fake_rccl_fail.argtypes = [ctypes.c_int, ctypes.c_int]
        fake.fake_rccl_fail.restype = None

        def stats():
            a = (ctypes.c_longlong * 8)()
            fake.fake_rccl_stats(a)
            return dict(zip(("all_gathers", "broadcasts", "groups", "copies", "inplace", "bytes", "created", "destroyed"), list(a)))

        def dalloc(nbytes):
            p = ctypes.c_void_p()
            _lib.check(lib.pmx_device_alloc(0, ctypes.byref(p), max(nbytes, 16)))
            return p

        def dfree(*ps):
            for p in ps:
                lib.pmx_device_free(0, p)

        def upload(dst, arr, stream):
            arr = np.ascontiguousarray(arr)
            if arr.nbytes:
                _lib.check(lib.pmx_device_upload(0, dst, ctypes.c_void_p(arr.ctypes.data), arr.nbytes, ctypes.c_void_p(stream)))

        def download(shape, src, stream, dtype=np.uint64):
            arr = np.zeros(shape, dtype=dtype)
            if arr.nbytes:
                _lib.check(lib.pmx_device_download(0, ctypes.c_void_p(arr.ctypes.data), src, arr.nbytes, ctypes.c_void_p(stream)))
            _lib.check(lib.pmx_stream_synchronize(0, ctypes.c_void_p(stream)))
            return arr

        def scenario(name):
            def deco(fn):
                try:
                    detail = fn()
                    record(name, True, detail or "")
                except Exception as e:   # noqa: BLE001
                    record(name, False, repr(e) + "\n" + traceback.format_exc())
                return fn
            return deco

        # ---- group creation: the hook is what lets one device fill several slots --------------------------------------------
        @scenario("create_refuses_shared_device_without_the_hook")
        def _():
            with_hook_off = ctypes.c_void_p()
            c = S.poseidon.c_config(cfg)
            arr = (ctypes.c_int * world)(*([0] * world))
            rc = lib.pmx_mgpu_create(ctypes.byref(c), world, arr, ctypes.byref(with_hook_off))
            assert rc == _lib.PMX_ERR_ARG and not with_hook_off.value, rc
            _lib.check(lib.pmx_mgpu_test_shared_device(1))
            rc = lib.pmx_mgpu_create(ctypes.byref(c), world, None, ctypes.byref(with_hook_off))      # devices must be explicit
            assert rc == _lib.PMX_ERR_ARG, rc
            rc = lib.pmx_mgpu_create(ctypes.byref(c), _lib.MAX_LOCAL_DEVICES + 1, arr, ctypes.byref(with_hook_off))
            assert rc == _lib.PMX_ERR_ARG, rc
            bad = (ctypes.c_int * world)(*([0] * (world - 1) + [lib.pmx_device_count()]))
            rc = lib.pmx_mgpu_create(ctypes.byref(c), world, bad, ctypes.byref(with_hook_off))
            assert rc == _lib.PMX_ERR_ARG and b"out of range" in lib.pmx_last_error(), rc

        _lib.check(lib.pmx_mgpu_test_shared_device(1))
        g = mgpu.DeviceGroup.single_process(cfg, devices=[0] * world)
        streams = [g.stream(l) for l in range(world)]

        @scenario("info_reports_world_ranks")
        def _():
            info = g.info()
            assert info["world"] == info["n_local"] == info["comm_ranks"] == world, info
            assert info["first_rank"] == info["comm_first_rank"] == 0 and info["devices"] == [0] * world and info["width"] == T, info
            assert len(set(streams)) == world and all(streams), streams
            assert g.stream(world) == 0 and not lib.pmx_mgpu_ctx(g._h, world) and not lib.pmx_mgpu_ctx(g._h, -1)
            assert len({g.context(l)._h.value for l in range(world)}) == world       # one context per slot
            return json.dumps(info)

        # ---- host batches: fan_out with `world` slots = world - 1 worker threads -------------------------------------------
        @scenario("host_batch_fan_out")
        def _():
            sizes = [1, world - 1, world, world + 1, 1000, 50001]
            for n in sizes:                 # n < world leaves the last slots empty (count == 0: no work, no thread error)
                states = synth.random_elements(cfg.field, n * T, seed=0x5EED0060 + n).reshape(n, T, 4)
                got = g.permute_batch(states)
                assert np.array_equal(got, cr.permute_batch(states, threads=0)), f"permute_batch n={n}"
            n, in_len, out_len = 20003, 5, 2
            msgs = synth.random_elements(cfg.field, n * in_len, seed=0x5EED0061).reshape(n, in_len, 4)
            assert np.array_equal(g.hash_batch(msgs, in_len, out_len), cr.hash_batch(msgs, in_len, out_len, threads=0))
            assert lib.pmx_mgpu_permute_batch(g._h, None, 0) == _lib.PMX_OK
            assert lib.pmx_mgpu_permute_batch(g._h, None, 5) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_hash_batch(g._h, None, 3, None, 1, 5) == _lib.PMX_ERR_ARG
            pinned = S.pinned_empty((4097, T, 4))
            st = synth.random_elements(cfg.field, 4097 * T, seed=0x5EED0062).reshape(4097, T, 4)
            pinned[:] = st
            g.permute_batch_inplace(pinned)                                           # page-locked: every slot's pipelined path
            assert np.array_equal(pinned, cr.permute_batch(st, threads=0))
            return f"sizes {sizes}"

        @scenario("fan_out_failure_on_a_nonzero_slot_and_serial_fallback")
        def _():
            n = 3001
            states = synth.random_elements(cfg.field, n * T, seed=0x5EED0063).reshape(n, T, 4)
            want = cr.permute_batch(states, threads=0)
            last = world - 1
            try:
                _lib.check(lib.pmx_mgpu_test_fault(last, 0))                          # a WORKER thread fails
                try:
                    g.permute_batch(states)
                    raise AssertionError("no error")
                except S.PmxError as e:
                    assert e.code == _lib.PMX_ERR_HIP and f"(slot {last}): injected failure" in str(e), str(e)
                try:
                    g.hash_batch(states, T, 1)
                    raise AssertionError("no error")
                except S.PmxError as e:
                    assert "injected failure" in str(e)
                _lib.check(lib.pmx_mgpu_test_fault(-1, 1))                            # no thread can be started: serial
                assert np.array_equal(g.permute_batch(states), want)
                _lib.check(lib.pmx_mgpu_test_fault(1, 1))                             # a serial slot fails
                try:
                    g.permute_batch(states)
                    raise AssertionError("no error")
                except S.PmxError as e:
                    assert "(slot 1): injected failure" in str(e), str(e)
                _lib.check(lib.pmx_mgpu_test_fault(0, 0))                             # the calling thread's slot fails
                try:
                    g.permute_batch(states)
                    raise AssertionError("no error")
                except S.PmxError as e:
                    assert "(slot 0): injected failure" in str(e), str(e)
            finally:
                _lib.check(lib.pmx_mgpu_test_fault(-1, 0))
            assert np.array_equal(g.permute_batch(states), want)                      # the group is intact afterwards

        # ---- device-resident shards + the gather, equal and ragged ---------------------------------------------------------
        def gather_case(n_total, row=T, label=""):
            whole = synth.random_elements(cfg.field, n_total * row, seed=0x5EED0064 + n_total).reshape(n_total, row, 4)
            shards, alls = [], []
            for l in range(world):
                start, count = g.local_span(n_total, l)
                d = dalloc(count * row * 32)                     # EXACTLY the shard: the stand-in checks every range
                upload(d, whole[start:start + count], streams[l])
                a = dalloc(n_total * row * 32)
                upload(a, np.full((n_total, row, 4), 0xA5A5A5A5A5A5A5A5, dtype=np.uint64), streams[l])
                shards.append(d)
                alls.append(a)
            before = stats()
            if row == T:
                g.permute_shards_dev([s.value for s in shards], n_total)
                want = cr.permute_batch(whole, threads=0)
            else:
                want = whole
            g.all_gather_dev([s.value for s in shards], [a.value for a in alls], n_total, row)
            g.synchronize()
            after = stats()
            for l in range(world):
                got = download((n_total, row, 4), alls[l], streams[l])
                assert np.array_equal(got, want), f"{label}: gathered copy of rank {l} (n_total {n_total})"
            dfree(*shards, *alls)
            return {k: after[k] - before[k] for k in after}

        @scenario("gather_equal_shards_is_one_all_gather")
        def _():
            d = gather_case(world * 4096, label="equal")
            assert d["all_gathers"] == 1 and d["broadcasts"] == 0 and d["groups"] == 1, d
            assert d["copies"] == world * world and d["bytes"] == world * world * 4096 * T * 32, d
            return json.dumps(d)

        @scenario("gather_ragged_shards_is_one_broadcast_per_rank")
        def _():
            out = []
            for extra in sorted({1, world - 1}):
                n_total = world * 3000 + extra
                d = gather_case(n_total, label="ragged")
                assert d["all_gathers"] == 0 and d["broadcasts"] == world and d["groups"] == 1, d
                assert d["inplace"] == 0 and d["copies"] == world * world, d          # root's shard -> its own span: a copy too
                assert d["bytes"] == world * n_total * T * 32, d
                out.append(d)
            return json.dumps(out)

        @scenario("gather_with_empty_shards")
        def _():
            n_total = world - 1                                   # the last rank has nothing: its broadcast is skipped by all
            d = gather_case(n_total, label="fewer units than ranks")
            assert d["broadcasts"] == world - 1, d
            empty = (ctypes.c_void_p * world)()
            assert lib.pmx_mgpu_all_gather_dev(g._h, empty, empty, 0, T) == _lib.PMX_OK
            assert lib.pmx_mgpu_all_gather_dev(g._h, empty, empty, 5, 0) == _lib.PMX_OK
            assert lib.pmx_mgpu_all_gather_dev(g._h, None, empty, 5, T) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_all_gather_dev(g._h, empty, empty, 1 << 62, 1 << 10) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_permute_shards_dev(g._h, None, 5) == _lib.PMX_ERR_ARG
            return json.dumps(d)

        @scenario("gather_of_digests_row_elems_1_and_2")
        def _():
            gather_case(world * 100 + 1, row=1, label="digests")
            gather_case(world * 64, row=2, label="two-element rows")

        @scenario("gather_in_place_slot")
        def _():
            """Equal shards with d_shards[l] = d_all[l] + l * shard: RCCL's in-place all-gather; nothing may be copied onto
            itself and every other slot must arrive."""
            per = 2048
            n_total = world * per
            whole = synth.random_elements(cfg.field, n_total * T, seed=0x5EED0066).reshape(n_total, T, 4)
            alls = []
            for l in range(world):
                a = dalloc(n_total * T * 32)
                upload(a, np.zeros((n_total, T, 4), dtype=np.uint64), streams[l])
                lib.pmx_device_upload(0, ctypes.c_void_p(a.value + l * per * T * 32), ctypes.c_void_p(whole[l * per:(l + 1) * per].ctypes.data),
                                      per * T * 32, ctypes.c_void_p(streams[l]))
                alls.append(a)
            shards = [a.value + l * per * T * 32 for l, a in enumerate(alls)]
            before = stats()
            g.permute_shards_dev(shards, n_total)
            g.all_gather_dev(shards, [a.value for a in alls], n_total, T)
            g.synchronize()
            d = {k: stats()[k] - before[k] for k in before}
            assert d["inplace"] == world and d["copies"] == world * (world - 1), d
            want = cr.permute_batch(whole, threads=0)
            for l in range(world):
                assert np.array_equal(download((n_total, T, 4), alls[l], streams[l]), want), f"rank {l}"
            dfree(*alls)
            return json.dumps(d)

        # ---- the gather to one rank, and the last step with its gather piece by piece (grouped ncclSend / ncclRecv) ----------
        fake.fake_rccl_messages.restype = ctypes.c_longlong

        def root_case(n_total, root, chunks=None, label=""):
            """chunks None: permute_shards_dev + gather_dev(root); else permute_gather_dev(root, chunks) (root < 0: every rank).
            Slots that do not receive get a NULL d_all; the receivers' copies must be the whole permuted batch, and - gather_dev -
            nobody else's buffers are touched."""
            whole = synth.random_elements(cfg.field, n_total * T, seed=0x5EED0070 + n_total + 31 * (root + 1)).reshape(n_total, T, 4)
            want = cr.permute_batch(whole, threads=0)
            receivers = list(range(world)) if root < 0 else [root]
            shards, alls = [], []
            for l in range(world):
                start, count = g.local_span(n_total, l)
                d = dalloc(count * T * 32)
                upload(d, whole[start:start + count], streams[l])
                shards.append(d)
                if l in receivers:
                    a = dalloc(n_total * T * 32)
                    upload(a, np.full((n_total, T, 4), 0xA5A5A5A5A5A5A5A5, dtype=np.uint64), streams[l])
                    alls.append(a)
                else:
                    alls.append(ctypes.c_void_p(0))
            before, m0 = stats(), fake.fake_rccl_messages()
            ptrs = [a.value or 0 for a in alls]
            if chunks is None:
                g.permute_shards_dev([s.value for s in shards], n_total)
                g.gather_dev([s.value for s in shards], ptrs, n_total, T, root)
            else:
                g.permute_gather_dev([s.value for s in shards], ptrs, n_total, root, chunks)
            # (no synchronize: the downloads below run on the slots' own streams, which must have waited for the transfers)
            for l in receivers:
                got = download((n_total, T, 4), alls[l], streams[l])
                assert np.array_equal(got, want), f"{label}: copy of rank {l} (n_total {n_total}, root {root}, chunks {chunks})"
            for l in range(world):                                  # every shard holds its own permuted span
                start, count = g.local_span(n_total, l)
                assert np.array_equal(download((count, T, 4), shards[l], streams[l]), want[start:start + count]), f"{label}: shard {l}"
            g.synchronize()
            d = {k: stats()[k] - before[k] for k in before}
            d["messages"] = fake.fake_rccl_messages() - m0
            dfree(*shards, *[a for a in alls if a.value])
            return d

        @scenario("gather_to_one_rank_is_world_minus_one_messages")
        def _():
            out = []
            for root in sorted({0, world - 1}):
                for n_total in (world * 2048, world * 1500 + 1, world * 1500 + world - 1):
                    d = root_case(n_total, root, label="to one rank")
                    per = [g.local_span(n_total, l)[1] for l in range(world)]
                    assert d["messages"] == world - 1 and d["groups"] == 1 and d["all_gathers"] == 0 and d["broadcasts"] == 0, d
                    assert d["bytes"] == (n_total - per[root]) * T * 32, d      # 1 / world of the all-gather's bytes per link
                    out.append(d)
            return json.dumps(out[-1])

        @scenario("gather_to_one_rank_with_empty_shards_and_bad_arguments")
        def _():
            d = root_case(world - 1, 0, label="fewer units than ranks")     # the last rank sends nothing, the root posts no receive for it
            assert d["messages"] == world - 2, d
            d = root_case(world - 1, world - 1, label="the root's own shard is the empty one")
            assert d["messages"] == world - 1, d
            empty = (ctypes.c_void_p * world)()
            some = dalloc(64)
            full = (ctypes.c_void_p * world)(*([some.value] * world))
            assert lib.pmx_mgpu_gather_dev(g._h, full, full, 0, T, 0) == _lib.PMX_OK
            assert lib.pmx_mgpu_gather_dev(g._h, full, full, 5, 0, 0) == _lib.PMX_OK
            assert lib.pmx_mgpu_gather_dev(g._h, None, full, 5, T, 0) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_gather_dev(g._h, full, full, 5, T, world) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_gather_dev(g._h, full, full, 5, T, -1) == _lib.PMX_ERR_ARG       # (every rank: pmx_mgpu_all_gather_dev)
            assert lib.pmx_mgpu_gather_dev(g._h, full, empty, 5, T, 0) == _lib.PMX_ERR_ARG and b"no buffer" in lib.pmx_last_error()
            assert lib.pmx_mgpu_gather_dev(g._h, full, full, 1 << 62, 1 << 10, 0) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_permute_gather_dev(g._h, full, full, 5, 0, 0) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_permute_gather_dev(g._h, full, full, 5, 0, 17) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_permute_gather_dev(g._h, full, full, 5, world, 2) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_permute_gather_dev(g._h, full, empty, 5, -1, 2) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_permute_gather_dev(g._h, full, full, 0, -1, 2) == _lib.PMX_OK
            dfree(some)

        @scenario("last_step_with_its_gather_piece_by_piece")
        def _():
            out = []
            for root in (-1, 0, world - 1):
                for n_total, chunks in ((world * 4096, 4), (world * 3000 + 1, 8), (world * 1000 + world - 1, 3), (world * 2 + 1, 16), (world * 4096, 1)):
                    d = root_case(n_total, root, chunks=chunks, label="piece by piece")
                    per = [g.local_span(n_total, l)[1] for l in range(world)]
                    pieces = lambda c: sum(1 for i in range(chunks) if (c // chunks) * (i + 1) + (c % chunks) * (i + 1) // chunks > (c // chunks) * i + (c % chunks) * i // chunks)
                    if root < 0:
                        assert d["messages"] == sum(pieces(c) for c in per) * (world - 1), (d, per, chunks)
                        assert d["bytes"] == (world - 1) * n_total * T * 32, d       # (a rank's own span is a copy on its device, not a message)
                    else:
                        assert d["messages"] == sum(pieces(c) for l, c in enumerate(per) if l != root), (d, per, chunks)
                    assert d["groups"] == chunks and d["all_gathers"] == 0 and d["broadcasts"] == 0, d
                    out.append(d)
            return json.dumps(out[:2])

        @scenario("wide_states_gathered_piece_by_piece_and_to_one_rank")
        def _():
            """t = 9 (rows of 36 words, the window engine of two waves per SIMD): a second group on the same slots, ragged shards, both new entry points"""
            name9, t9 = "bn254_t9_a5_8_57", 9
            cfg9, cr9 = product_config(name9), c_oracle(name9)
            g9 = mgpu.DeviceGroup.single_process(cfg9, devices=[0] * world)
            try:
                st9 = [g9.stream(l) for l in range(world)]
                n_total = world * 700 + 1
                whole = synth.random_elements(cfg9.field, n_total * t9, seed=0x5EED0079).reshape(n_total, t9, 4)
                want = cr9.permute_batch(whole, threads=0)
                for root, chunks in ((-1, 3), (world - 1, None), (0, 5)):
                    receivers = list(range(world)) if root < 0 else [root]
                    shards, alls = [], []
                    for l in range(world):
                        start, count = g9.local_span(n_total, l)
                        d = dalloc(count * t9 * 32)
                        upload(d, whole[start:start + count], st9[l])
                        shards.append(d)
                        if l in receivers:
                            a = dalloc(n_total * t9 * 32)
                            upload(a, np.zeros((n_total, t9, 4), dtype=np.uint64), st9[l])
                            alls.append(a)
                        else:
                            alls.append(ctypes.c_void_p(0))
                    ptrs = [a.value or 0 for a in alls]
                    if chunks is None:
                        g9.permute_shards_dev([x.value for x in shards], n_total)
                        g9.gather_dev([x.value for x in shards], ptrs, n_total, t9, root)
                    else:
                        g9.permute_gather_dev([x.value for x in shards], ptrs, n_total, root, chunks)
                    for l in receivers:
                        assert np.array_equal(download((n_total, t9, 4), alls[l], st9[l]), want), (root, chunks, l)
                    g9.synchronize()
                    dfree(*shards, *[a for a in alls if a.value])
            finally:
                g9.close()

        @scenario("stand_in_refuses_a_send_nobody_receives")
        def _():
            """The checker checks: a failure injected into the root's first ncclRecv voids the group (nothing hangs, the error comes back as
            PMX_ERR_RCCL), and the group works afterwards."""
            n_total = world * 64
            shards = [dalloc(g.local_span(n_total, l)[1] * T * 32) for l in range(world)]
            a = dalloc(n_total * T * 32)
            alls = [a.value] + [0] * (world - 1)
            fake.fake_rccl_fail(9, 1)
            rc = lib.pmx_mgpu_gather_dev(g._h, mgpu._ptr_array([s.value for s in shards]), mgpu._ptr_array(alls), n_total, T, 0)
            msg = lib.pmx_last_error().decode()
            fake.fake_rccl_fail(0, 0)
            assert rc == _lib.PMX_ERR_RCCL and "ncclRecv" in msg, (rc, msg)
            fake.fake_rccl_fail(8, 1)
            rc = lib.pmx_mgpu_permute_gather_dev(g._h, mgpu._ptr_array([s.value for s in shards]), mgpu._ptr_array([a.value] * world), n_total, -1, 2)
            msg = lib.pmx_last_error().decode()
            fake.fake_rccl_fail(0, 0)
            assert rc == _lib.PMX_ERR_RCCL and "ncclSend" in msg, (rc, msg)
            g.synchronize()
            dfree(*shards, a)
            root_case(world * 256, 0, label="after the refused groups")
            return msg

        @scenario("stand_in_catches_a_gather_buffer_that_is_too_small")
        def _():
            """The checker checks: a d_all one row short must be refused by the stand-in (PMX_ERR_RCCL, range message)."""
            n_total = world * 10 + 1
            shards = [dalloc(g.local_span(n_total, l)[1] * T * 32) for l in range(world)]
            alls = [dalloc((n_total - (1 if l == world - 1 else 0)) * T * 32) for l in range(world)]
            rc = lib.pmx_mgpu_all_gather_dev(g._h, mgpu._ptr_array([s.value for s in shards]), mgpu._ptr_array([a.value for a in alls]), n_total, T)
            msg = lib.pmx_last_error().decode()
            g.synchronize()
            dfree(*shards, *alls)
            assert rc == _lib.PMX_ERR_RCCL and "of an allocation of" in msg, (rc, msg)
            return msg

        @scenario("hash_driver_per_shard_then_gather")
        def _():
            n_total, in_len, out_len = 30000 + 7, 5, 2
            msgs = synth.random_elements(cfg.field, n_total * in_len, seed=0x5EED0067).reshape(n_total, in_len, 4)
            d_in, d_out, d_all = [], [], []
            for l in range(world):
                start, count = g.local_span(n_total, l)
                i, o, a = dalloc(count * in_len * 32), dalloc(count * out_len * 32), dalloc(n_total * out_len * 32)
                upload(i, msgs[start:start + count], streams[l])
                g.context(l).hash_batch_dev(i.value, in_len, o.value, out_len, count, streams[l])
                d_in.append(i), d_out.append(o), d_all.append(a)
            g.all_gather_dev([o.value for o in d_out], [a.value for a in d_all], n_total, out_len)
            g.synchronize()
            want = cr.hash_batch(msgs, in_len, out_len, threads=0)
            for l in range(world):
                assert np.array_equal(download((n_total, out_len, 4), d_all[l], streams[l]), want), f"rank {l}"
            dfree(*d_in, *d_out, *d_all)

        # ---- the sharded tree ----------------------------------------------------------------------------------------------
        pow2 = world & (world - 1) == 0

        @scenario("sharded_merkle_host_leaves")
        def _():
            if not pow2:
                rc = lib.pmx_mgpu_merkle_2to1(g._h, ctypes.c_void_p(np.zeros((world * 2, 4), dtype=np.uint64).ctypes.data), world * 2,
                                              ctypes.c_void_p(np.zeros(4, dtype=np.uint64).ctypes.data))
                assert rc == _lib.PMX_ERR_ARG and b"power-of-two number of ranks" in lib.pmx_last_error()
                return "world is not a power of two: refused"
            for log2_m in (0, 1, 7, 12):
                m = world << log2_m
                leaves = synth.random_elements(cfg.field, m, seed=0x5EED0068 + log2_m)
                assert np.array_equal(g.merkle_root(leaves), cr.merkle(leaves, threads=0)[-1]), f"root of {m} leaves"
            for bad in (0, world // 2, 3 * world):
                rc = lib.pmx_mgpu_merkle_2to1(g._h, ctypes.c_void_p(np.zeros((max(bad, 1), 4), dtype=np.uint64).ctypes.data), bad,
                                              ctypes.c_void_p(np.zeros(4, dtype=np.uint64).ctypes.data))
                assert rc == _lib.PMX_ERR_ARG, (bad, rc)
            assert lib.pmx_mgpu_merkle_2to1(g._h, None, 8, None) == _lib.PMX_ERR_ARG

        @scenario("sharded_merkle_device_resident_every_rank_holds_the_whole_top")
        def _():
            if not pow2:
                empty = (ctypes.c_void_p * world)()
                assert lib.pmx_mgpu_merkle_2to1_dev(g._h, empty, empty, world * 4) == _lib.PMX_ERR_ARG
                return "world is not a power of two: refused"
            m = 1 << 9
            n_leaves = world * m
            leaves = synth.random_elements(cfg.field, n_leaves, seed=0x5EED0069)
            want = cr.merkle(leaves, threads=0)                                 # [2 n - 1][4]: leaves, every level, root last
            d_nodes, d_top = [], []
            for l in range(world):
                nd, tp = dalloc((2 * m - 1) * 32), dalloc((2 * world - 1) * 32)           # exact sizes
                upload(nd, leaves[l * m:(l + 1) * m], streams[l])
                d_nodes.append(nd), d_top.append(tp)
            g.merkle_2to1_dev([d.value for d in d_nodes], [d.value for d in d_top], n_leaves)
            g.synchronize()
            # the top 2W - 1 nodes of the whole tree are its last 2W - 1 rows: W subtree roots, then the levels above
            want_top = want[-(2 * world - 1):]
            for l in range(world):
                top = download((2 * world - 1, 4), d_top[l], streams[l])
                assert np.array_equal(top, want_top), f"d_top of rank {l}"
                sub = download((2 * m - 1, 4), d_nodes[l], streams[l])
                assert np.array_equal(sub[-1], want_top[l]), f"subtree root of rank {l}"
                assert np.array_equal(sub, cr.merkle(leaves[l * m:(l + 1) * m], threads=0)), f"subtree of rank {l}"
            dfree(*d_nodes, *d_top)
            empty = (ctypes.c_void_p * world)()
            assert lib.pmx_mgpu_merkle_2to1_dev(g._h, None, empty, n_leaves) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_merkle_2to1_dev(g._h, empty, empty, world // 2) == _lib.PMX_ERR_ARG

        # ---- RCCL failures come back as PMX_ERR_RCCL with the library's text ----------------------------------------------------
        @scenario("collective_failures_are_status_codes")
        def _():
            n_total = world * 8
            shards = [dalloc(8 * T * 32) for _ in range(world)]
            alls = [dalloc(n_total * T * 32) for _ in range(world)]
            sp, ap = mgpu._ptr_array([s.value for s in shards]), mgpu._ptr_array([a.value for a in alls])
            seen = []
            for which, n_units, needle in ((1, n_total, "ncclAllGather"), (7, n_total, "GroupStart"), (3, n_total, "ncclGroupEnd"),
                                           (2, n_total - 1, "ncclBroadcast")):
                fake.fake_rccl_fail(which, 1 if which != 1 else world)        # the LAST slot's all-gather call fails
                rc = lib.pmx_mgpu_all_gather_dev(g._h, sp, ap, n_units, T)
                msg = lib.pmx_last_error().decode()
                fake.fake_rccl_fail(0, 0)
                assert rc == _lib.PMX_ERR_RCCL and needle in msg and "injected failure" in msg, (which, rc, msg)
                seen.append(msg)
            if pow2:
                nodes = [dalloc(32) for _ in range(world)]
                tops = [dalloc((2 * world - 1) * 32) for _ in range(world)]
                for which, needle in ((1, "ncclAllGather"), (3, "ncclGroupEnd"), (7, "GroupStart")):
                    fake.fake_rccl_fail(which, 1)
                    rc = lib.pmx_mgpu_merkle_2to1_dev(g._h, mgpu._ptr_array([d.value for d in nodes]), mgpu._ptr_array([d.value for d in tops]), world)
                    msg = lib.pmx_last_error().decode()
                    fake.fake_rccl_fail(0, 0)
                    assert rc == _lib.PMX_ERR_RCCL and needle in msg, (which, rc, msg)
                g.synchronize()
                dfree(*nodes, *tops)
            g.synchronize()
            g.all_gather_dev([s.value for s in shards], [a.value for a in alls], n_total, T)      # and the group still works
            g.synchronize()
            dfree(*shards, *alls)
            c = S.poseidon.c_config(cfg)
            h = ctypes.c_void_p()
            arr = (ctypes.c_int * world)(*([0] * world))
            fake.fake_rccl_fail(4, 1)
            rc = lib.pmx_mgpu_create(ctypes.byref(c), world, arr, ctypes.byref(h))
            assert rc == _lib.PMX_ERR_RCCL and b"ncclCommInitAll" in lib.pmx_last_error() and not h.value
            fake.fake_rccl_fail(5, 1)
            rc = lib.pmx_mgpu_create_rank(ctypes.byref(c), 0, 0, 1, (ctypes.c_uint8 * 128)(), ctypes.byref(h))
            assert rc == _lib.PMX_ERR_RCCL and b"ncclCommInitRank" in lib.pmx_last_error() and not h.value
            fake.fake_rccl_fail(6, 1)
            rc = lib.pmx_mgpu_unique_id((ctypes.c_uint8 * 128)())
            assert rc == _lib.PMX_ERR_RCCL and b"GetUniqueId" in lib.pmx_last_error()
            assert lib.pmx_mgpu_unique_id(None) == _lib.PMX_ERR_ARG
            return json.dumps(seen)

        # ---- one rank per THREAD with ncclCommInitRank: groups whose first_rank is not 0 ----------------------------------------
        @scenario("one_group_per_rank_create_rank")
        def _():
            """What bench.py --gpus N and a one-process-per-GPU Rust job do, with threads standing in for the processes: every
            rank makes its own one-slot group (pmx_mgpu_create_rank: first_rank = rank, n_local = 1, world = W) from one id and
            drives its shard; gather (equal and ragged) and the sharded tree.  The stand-in joins the ranks by id and
            rendezvouses every collective across the threads."""
            uid = mgpu.unique_id()
            n_equal, n_ragged, m = world * 1000, world * 1000 + (world - 1), 1 << 6
            whole = {n: synth.random_elements(cfg.field, n * T, seed=0x5EED0070 + n).reshape(n, T, 4) for n in (n_equal, n_ragged)}
            want = {n: cr.permute_batch(w, threads=0) for n, w in whole.items()}
            leaves = synth.random_elements(cfg.field, world * m, seed=0x5EED0071)
            want_tree = cr.merkle(leaves, threads=0) if pow2 else None
            errors, infos = [], [None] * world

            def rank_main(r):
                try:
                    gr = mgpu.DeviceGroup.one_rank(cfg, 0, r, world, uid)
                    info = gr.info()
                    infos[r] = info
                    assert info["world"] == world and info["n_local"] == 1 and info["first_rank"] == r, info
                    assert info["comm_ranks"] == world and info["comm_first_rank"] == r, info
                    st = gr.stream(0)
                    for n_total in (n_equal, n_ragged):
                        start, count = gr.local_span(n_total, 0)
                        assert (start, count) == mgpu.shard_bounds(n_total, world, r)
                        d, a = dalloc(count * T * 32), dalloc(n_total * T * 32)
                        upload(d, whole[n_total][start:start + count], st)
                        gr.permute_shards_dev([d.value], n_total)
                        gr.all_gather_dev([d.value], [a.value], n_total, T)
                        got = download((n_total, T, 4), a, st)
                        assert np.array_equal(got, want[n_total]), f"rank {r}: gathered copy, n_total {n_total}"
                        dfree(d, a)
                    if pow2:
                        nd, tp = dalloc((2 * m - 1) * 32), dalloc((2 * world - 1) * 32)
                        upload(nd, leaves[r * m:(r + 1) * m], st)
                        gr.merkle_2to1_dev([nd.value], [tp.value], world * m)
                        top = download((2 * world - 1, 4), tp, st)
                        assert np.array_equal(top, want_tree[-(2 * world - 1):]), f"rank {r}: top of the tree"
                        dfree(nd, tp)
                    # the host-batch entry points need all ranks in one process group: a one-rank slice must say so
                    one = np.zeros((4, T, 4), dtype=np.uint64)
                    rc = lib.pmx_mgpu_permute_batch(gr._h, ctypes.c_void_p(one.ctypes.data), 4)
                    assert rc == _lib.PMX_ERR_ARG and b"single-process group" in lib.pmx_last_error()
                    rc = lib.pmx_mgpu_merkle_2to1(gr._h, ctypes.c_void_p(one.ctypes.data), 4, ctypes.c_void_p(one.ctypes.data))
                    assert rc == _lib.PMX_ERR_ARG and b"single-process group" in lib.pmx_last_error()
                    gr.synchronize()
                    gr.close()
                except Exception as e:   # noqa: BLE001
                    errors.append(f"rank {r}: {e!r}\n{traceback.format_exc()[-800:]}")

            threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
            for th in threads:
                th.start()
            for th in threads:
                th.join(600)
            assert not any(th.is_alive() for th in threads), "a rank is stuck"
            assert not errors, "\n".join(errors)
            return json.dumps([i["comm_first_rank"] for i in infos])

        @scenario("create_rank_argument_errors")
        def _():
            c = S.poseidon.c_config(cfg)
            h = ctypes.c_void_p()
            uid = (ctypes.c_uint8 * 128)()
            for args in ((0, world, world), (0, -1, world), (0, 0, 0), (lib.pmx_device_count(), 0, world), (-1, 0, world)):
                rc = lib.pmx_mgpu_create_rank(ctypes.byref(c), args[0], args[1], args[2], uid, ctypes.byref(h))
                assert rc == _lib.PMX_ERR_ARG and not h.value, (args, rc)
            assert lib.pmx_mgpu_create_rank(None, 0, 0, 1, uid, ctypes.byref(h)) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_create_rank(ctypes.byref(c), 0, 0, 1, None, ctypes.byref(h)) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_create(None, 1, None, None) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_create(ctypes.byref(c), 0, None, ctypes.byref(h)) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_get_info(None, None) == _lib.PMX_ERR_ARG and lib.pmx_mgpu_synchronize(None) == _lib.PMX_ERR_ARG
            assert lib.pmx_mgpu_destroy(None) == _lib.PMX_OK
            bad_cfg = S.poseidon.c_config(cfg)
            bad_cfg.rate = 0                                            # group_contexts fails: the half-built group is torn down
            arr = (ctypes.c_int * world)(*([0] * world))
            assert lib.pmx_mgpu_create(ctypes.byref(bad_cfg), world, arr, ctypes.byref(h)) == _lib.PMX_ERR_CONFIG and not h.value
            assert lib.pmx_mgpu_create_rank(ctypes.byref(bad_cfg), 0, 0, 1, uid, ctypes.byref(h)) == _lib.PMX_ERR_CONFIG

        @scenario("shard_bounds_argument_errors")
        def _():
            s, c = ctypes.c_size_t(), ctypes.c_size_t()
            assert lib.pmx_shard_bounds(10, 0, 0, ctypes.byref(s), ctypes.byref(c)) == _lib.PMX_ERR_ARG
            assert lib.pmx_shard_bounds(10, 2, 2, ctypes.byref(s), ctypes.byref(c)) == _lib.PMX_ERR_ARG
            assert lib.pmx_shard_bounds(10, 2, 0, None, ctypes.byref(c)) == _lib.PMX_ERR_ARG

        g.close()

        @scenario("world_of_one_behind_the_stand_in")
        def _():
            """The W == 1 shortcuts (root copied straight into d_top, a one-rank all-gather) on the same bookkeeping."""
            g1 = mgpu.DeviceGroup.single_process(cfg, devices=[0])
            leaves = synth.random_elements(cfg.field, 64, seed=0x5EED0072)
            want = cr.merkle(leaves, threads=0)
            assert np.array_equal(g1.merkle_root(leaves), want[-1])
            assert np.array_equal(g1.merkle_root(leaves[:1]), leaves[0])
            n = 777
            whole = synth.random_elements(cfg.field, n * T, seed=0x5EED0073).reshape(n, T, 4)
            d, a = dalloc(n * T * 32), dalloc(n * T * 32)
            upload(d, whole, g1.stream(0))
            g1.permute_shards_dev([d.value], n)
            g1.all_gather_dev([d.value], [a.value], n, T)
            assert np.array_equal(download((n, T, 4), a, g1.stream(0)), cr.permute_batch(whole, threads=0))
            dfree(d, a)
            g1.close()

        _lib.check(lib.pmx_mgpu_test_shared_device(0))
        final = stats()
        res["fake_rccl_stats"] = final
        record("every_communicator_was_destroyed", final["created"] == final["destroyed"] and final["created"] > 0, json.dumps(final))
        res["ok"] = all(s["ok"] for s in res["scenarios"])
    except Exception as e:       # noqa: BLE001
        res["error"] = repr(e) + "\n" + traceback.format_exc()[-2500:]
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
    sys.exit(0 if res["ok"] else 1)


if __name__ == "__main__":
    main()
