"""BASELINE.json's full sizes on the GPU: the whole batch against the C restatement (all host cores) where
that takes seconds, plus size-independent properties (position independence, 2-to-1 == permute of
[0,l,r], subtree decomposition of the Merkle root)."""
import numpy as np
import pytest
import torch

import sponge_amd as S
from sponge_amd import synth

from gpu_helpers import c_oracle, product_config

pytestmark = pytest.mark.gpu


def dev_tensor(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(a.view(np.int64)).to("cuda:0")


def to_numpy(t: torch.Tensor) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


def test_c2_full_batch_2e20_bls_t3_alpha5():
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    n = 1 << 20
    states = synth.random_elements(cfg.field, n * 3, seed=0x5EED0002).reshape(n, 3, 4)
    d = dev_tensor(states)
    ctx = cfg.context()
    ctx.permute_batch_dev(d.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = to_numpy(d).reshape(n, 3, 4)
    want = c_oracle(name).permute_batch(states, threads=0)
    assert np.array_equal(got, want)
    # position independence: the reversed batch gives the reversed result
    d2 = dev_tensor(np.ascontiguousarray(states[::-1]))
    ctx.permute_batch_dev(d2.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(to_numpy(d2).reshape(n, 3, 4)[::-1], got)


def test_c3_full_batch_2e18_bn254_t9_alpha5():
    name = "bn254_t9_a5_8_57"
    cfg = product_config(name)
    n = 1 << 18
    states = synth.random_elements(cfg.field, n * 9, seed=0x5EED0003).reshape(n, 9, 4)
    d = dev_tensor(states)
    cfg.context().permute_batch_dev(d.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = to_numpy(d).reshape(n, 9, 4)
    # the whole batch (the restatement needs ~0.25 ms per t=9 permutation per core: a few seconds on the box's cores)
    want = c_oracle(name).permute_batch(states, threads=0)
    assert np.array_equal(got, want)


def test_c3_wide_sponge_driver_2e18_hash_8_to_1_through_absorb_and_squeeze():
    """The absorb / squeeze batch driver at BASELINE configs[2]'s size: 2^18 fresh BN254 Fr t = 9 sponges, device-resident
    with explicit mode words, absorb(8) then squeeze_native(1) - `new; absorb; squeeze` per row, mod.rs:219-254, 321-341 -
    the whole batch against the C port's hash rows; then absorb(11) + squeeze(9) into the same sponges (now Squeezing{1}:
    every one permutes first) against the C port run on a sample sponge by sponge."""
    name = "bn254_t9_a5_8_57"
    cfg = product_config(name)
    cr = c_oracle(name)
    n, t, r = 1 << 18, 9, 8
    st = torch.zeros((n, t, 4), dtype=torch.int64, device="cuda:0")
    tag = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    idx = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    msgs = synth.random_elements(cfg.field, n * r, seed=0x5EED0009).reshape(n, r, 4)
    d_in = dev_tensor(msgs)
    d_out = torch.zeros((n, 1, 4), dtype=torch.int64, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    ctx = cfg.context()
    ctx.sponge_absorb_batch_dev(st.data_ptr(), tag.data_ptr(), idx.data_ptr(), d_in.data_ptr(), r, n, stream)
    torch.cuda.synchronize()
    assert int(tag.max()) == 0 and int(idx.min()) == int(idx.max()) == r      # the rate filled exactly: no permutation yet
    ctx.sponge_squeeze_batch_dev(st.data_ptr(), tag.data_ptr(), idx.data_ptr(), d_out.data_ptr(), 1, n, stream)
    torch.cuda.synchronize()
    assert np.array_equal(to_numpy(d_out).reshape(n, 1, 4), cr.hash_batch(msgs, r, 1, threads=0))
    assert int(tag.min()) == 1 and int(idx.min()) == int(idx.max()) == 1
    before = to_numpy(st).reshape(n, t, 4).copy()
    more = synth.random_elements(cfg.field, n * 11, seed=0x5EED000A).reshape(n, 11, 4)
    d_more = dev_tensor(more)
    d_out9 = torch.zeros((n, 9, 4), dtype=torch.int64, device="cuda:0")
    ctx.sponge_absorb_batch_dev(st.data_ptr(), tag.data_ptr(), idx.data_ptr(), d_more.data_ptr(), 11, n, stream)
    ctx.sponge_squeeze_batch_dev(st.data_ptr(), tag.data_ptr(), idx.data_ptr(), d_out9.data_ptr(), 9, n, stream)
    torch.cuda.synchronize()
    after, out9 = to_numpy(st).reshape(n, t, 4), to_numpy(d_out9).reshape(n, 9, 4)
    sample = np.unique(np.concatenate([np.arange(300), np.arange(n - 300, n), np.linspace(0, n - 1, 400).astype(np.int64)]))
    for j in sample:
        s, m, i = cr.sponge_absorb(before[j], 1, 1, more[j])
        s, m, i, o = cr.sponge_squeeze(s, m, i, 9)
        assert np.array_equal(after[j], s) and np.array_equal(out9[j], o), j
        assert (int(tag[j]), int(idx[j])) == (m, i), j


def test_c5_merkle_2e20_leaves_root_and_decomposition():
    """Level-by-level tree on the GPU; root == root rebuilt from 8 subtree roots (the multi-GPU split);
    lowest level == 2-to-1 compression == permute([0, l, r])[capacity] checked against the restatement."""
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    ctx = cfg.context()
    m = 1 << 20
    leaves = synth.random_elements(cfg.field, m, seed=0x5EED0005)
    nodes = torch.zeros((2 * m - 1, 4), dtype=torch.int64, device="cuda:0")
    nodes[:m] = dev_tensor(leaves)
    ctx.merkle_2to1_dev(nodes.data_ptr(), m, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = to_numpy(nodes)
    # first level against the restatement, in full
    want_l1 = c_oracle(name).hash_batch(leaves.reshape(m // 2, 2, 4), 2, 1, threads=0).reshape(m // 2, 4)
    assert np.array_equal(got[m:m + m // 2], want_l1)
    # 2-to-1 == permute([0,l,r])[1] on a sample
    st = np.zeros((1024, 3, 4), dtype=np.uint64)
    st[:, 1:, :] = leaves[:2048].reshape(1024, 2, 4)
    assert np.array_equal(ctx.permute_batch(st)[:, 1, :], got[m:m + 1024])
    # subtree decomposition: 8 shards of 2^17 leaves -> 8 roots -> 3 more levels
    sub_roots = []
    for g in range(8):
        _, r = ctx.merkle_2to1(leaves[g * (m // 8):(g + 1) * (m // 8)], want_nodes=False)
        sub_roots.append(r)
    _, top = ctx.merkle_2to1(np.stack(sub_roots), want_nodes=False)
    assert np.array_equal(top, got[-1])
    # upper 10 levels against the restatement (1023 compressions)
    lvl10 = got[2 * m - 1 - 2047: 2 * m - 1 - 1023]      # the level with 1024 nodes
    want_top = c_oracle(name).merkle(lvl10, threads=0)
    assert np.array_equal(want_top[-1], got[-1])


def test_wide_and_narrow_compression_kernels_agree_at_the_switch():
    """A 2^17-leaf tree: level 1 (65536 parents) runs on the window engine, level 2 (32768) and above on the cooperative
    kernel; both against the C restatement."""
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    m = 1 << 17
    leaves = synth.random_elements(cfg.field, m, seed=0x5EED0017)
    nodes, root = cfg.context().merkle_2to1(leaves)
    want = c_oracle(name).merkle(leaves, threads=0)
    assert np.array_equal(nodes, want)


@pytest.mark.parametrize("alpha,rf,rp", [(5, 8, 31), (17, 8, 31), (257, 8, 13)])
def test_every_large_batch_kernel_of_t3_vs_c_oracle(alpha, rf, rp):
    """t = 3 launches that fill the device (2^17 units and more; the small-batch tests stay below): every entry point for the
    exponent with a dedicated chain and two on the generic S-box, against the C port: permute, hash, mid-stream absorb +
    squeeze with mixed modes, and a Merkle tree whose widest level has 2^18 compressions - all on the window engine of t = 3."""
    from oracle import cref
    from oracle import poseidon_oracle as O
    f = S.BLS12_381_FR
    cfg = S.poseidon_config_from_lfsr(f, 2, alpha, rf, rp)
    cr = cref.CRef(O.make_config(O.BLS12_381_FR, 255, 2, alpha, rf, rp))
    ctx = cfg.context()
    n = (1 << 17) + 77                                   # ragged on purpose
    states = synth.random_elements(f, n * 3, seed=alpha).reshape(n, 3, 4)
    assert np.array_equal(ctx.permute_batch(states), cr.permute_batch(states, threads=0))
    msgs = synth.random_elements(f, n * 3, seed=alpha + 1).reshape(n, 3, 4)
    assert np.array_equal(ctx.hash_batch(msgs, 3, 2), cr.hash_batch(msgs, 3, 2, threads=0))
    # mid-stream driver: every sponge absorbs 3 elements, then squeezes 3; the first half starts mid-absorb
    b = S.BatchPoseidonSponge.new(cfg, n)
    b.state[:] = states
    b.mode_index[: n // 2] = 1
    b.mode_tag[n // 2:: 3] = 1                           # PMX_MODE_SQUEEZING (index 0)
    st0, tag0, idx0 = b.state.copy(), b.mode_tag.copy(), b.mode_index.copy()
    b.absorb(msgs)
    got = b.squeeze_native_field_elements(3)
    picks = np.concatenate([np.arange(0, 4096), np.arange(n // 2 - 2048, n // 2 + 2048), np.arange(n - 4096, n)])
    for i in picks[:: 7]:
        s, m, x = cr.sponge_absorb(st0[i], int(tag0[i]), int(idx0[i]), msgs[i])
        s, m, x, out = cr.sponge_squeeze(s, m, x, 3)
        assert np.array_equal(got[i], out) and np.array_equal(b.state[i], s) and (int(b.mode_tag[i]), int(b.mode_index[i])) == (m, x), i
    leaves = synth.random_elements(f, 1 << 19, seed=alpha + 2)
    nodes, root = ctx.merkle_2to1(leaves)
    want = cr.merkle(leaves, threads=0)
    assert np.array_equal(nodes, want) and np.array_equal(root, want[-1])


# ---- the per-GPU shards of the 8-GPU configurations, and the whole C5 tree on one GPU ---------------------------------
def test_c4_per_gpu_shard_2e21_states_whole_batch():
    """BASELINE configs[3] puts 2^24 states on 8 GPUs: 2^21 per GPU.  Shard 5 of that batch (the global seeded batch at
    offset 5 * 2^21, exactly what bench.py gives rank 5), whole shard against the C restatement."""
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    n, shard = 1 << 21, 5
    states = synth.random_elements(cfg.field, n * 3, seed=0x5EED0002, offset=shard * n * 3).reshape(n, 3, 4)
    d = dev_tensor(states)
    cfg.context().permute_batch_dev(d.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(to_numpy(d).reshape(n, 3, 4), c_oracle(name).permute_batch(states, threads=0))


def test_c5_per_gpu_subtree_2e21_leaves_all_nodes():
    """BASELINE configs[4] on 8 GPUs: every GPU reduces a 2^21-leaf subtree.  All 2^22 - 1 nodes against the C restatement."""
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    m = 1 << 21
    leaves = synth.random_elements(cfg.field, m, seed=0x5EED0005, offset=3 * m)        # rank 3's leaves
    nodes = torch.zeros((2 * m - 1, 4), dtype=torch.int64, device="cuda:0")
    nodes[:m] = dev_tensor(leaves)
    cfg.context().merkle_2to1_dev(nodes.data_ptr(), m, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(to_numpy(nodes), c_oracle(name).merkle(leaves, threads=0))


def test_c5_whole_tree_2e24_leaves_on_one_gpu():
    """The whole C5 tree - 2^24 leaves, 2^24 - 1 compressions, 1 GiB of nodes - on one GPU: every node against the C
    restatement, and the root against the root rebuilt from the 8 subtree roots (the 8-GPU decomposition: subtree g is
    leaves [g 2^21, (g+1) 2^21), its root is node g of the level that has 8 nodes)."""
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    ctx = cfg.context()
    m = 1 << 24
    leaves = synth.random_elements(cfg.field, m, seed=0x5EED0005)
    nodes = torch.zeros((2 * m - 1, 4), dtype=torch.int64, device="cuda:0")
    nodes[:m] = dev_tensor(leaves)
    ctx.merkle_2to1_dev(nodes.data_ptr(), m, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = to_numpy(nodes)
    del nodes
    # the 8-GPU split: each 2^21-leaf subtree on its own, then 3 more levels over the 8 roots
    sub_roots = []
    for g in (0, 7):                                  # two of the eight as separate launches ...
        sub = torch.zeros((2 * (m // 8) - 1, 4), dtype=torch.int64, device="cuda:0")
        sub[:m // 8] = dev_tensor(leaves[g * (m // 8):(g + 1) * (m // 8)])
        ctx.merkle_2to1_dev(sub.data_ptr(), m // 8, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        sub_roots.append((g, to_numpy(sub[-1:]).reshape(4)))
        del sub
    level8 = got[2 * m - 16: 2 * m - 8]               # ... all eight are the level with 8 nodes of the big tree
    for g, r in sub_roots:
        assert np.array_equal(level8[g], r)
    _, top = ctx.merkle_2to1(level8, want_nodes=False)
    assert np.array_equal(top, got[-1])
    want = c_oracle(name).merkle(leaves, threads=0)
    assert np.array_equal(got, want)


def test_c4_whole_batch_2e24_states_through_the_device_group():
    """BASELINE configs[3] at its total size in ONE batch: 2^24 states (1.5 GiB) cut into world = pmx_device_count()
    contiguous shards, pmx_mgpu_permute_shards_dev on the group's streams, pmx_mgpu_all_gather_dev (RCCL) into one
    buffer per device, and that gathered buffer against the C restatement IN FULL (16 M permutations on the host cores).
    One GPU: one 2^24-state launch and a one-rank ncclAllGather; N GPUs: the real C4."""
    from sponge_amd import _lib, mgpu
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    g = mgpu.DeviceGroup.single_process(cfg, _lib.lib().pmx_device_count())
    n_total = 1 << 24
    whole = synth.random_elements(cfg.field, n_total * 3, seed=0x5EED0002).reshape(n_total, 3, 4)
    shards, alls = [], []
    for l, dev in enumerate(g.devices):
        start, count = g.local_span(n_total, l)
        shards.append(torch.from_numpy(whole[start:start + count].view(np.int64)).to(f"cuda:{dev}"))
        alls.append(torch.zeros((n_total, 3, 4), dtype=torch.int64, device=f"cuda:{dev}"))
    for dev in g.devices:
        torch.cuda.synchronize(dev)
    g.permute_shards_dev([s.data_ptr() for s in shards], n_total)
    g.all_gather_dev([s.data_ptr() for s in shards], [a.data_ptr() for a in alls], n_total, 3)
    g.synchronize()
    cr = c_oracle(name)
    chunk = 1 << 21
    for first in range(0, n_total, chunk):          # the checker in slices: bounded host memory
        want = cr.permute_batch(whole[first:first + chunk], threads=0)
        for l in range(g.n_local):
            if l == 0 or first % (4 * chunk) == 0:  # device 0's copy in full, the other copies on a quarter of the slices
                got = alls[l][first:first + chunk].cpu().numpy().view(np.uint64)
                assert np.array_equal(got, want), (l, first)
    g.close()


def test_forest_of_2e11_trees_is_the_lower_half_of_one_2e21_leaf_tree():
    """Size-independent property at the per-GPU size of BASELINE configs[4]: 2^11 trees of 2^10 leaves advanced together
    (pmx_merkle_2to1_forest_dev) produce, level by level, exactly the first ten levels of the ONE tree over the same 2^21
    leaves (pmx_merkle_2to1_dev) - pairs never straddle trees -, and a sample of the trees against the C port."""
    name = "bls_t3_a5_8_31"
    cfg = product_config(name)
    n_trees, m = 1 << 11, 1 << 10
    total = n_trees * m
    leaves = synth.random_elements(cfg.field, total, seed=0x5EED00F5)
    ctx = cfg.context()
    stream = torch.cuda.current_stream().cuda_stream
    one = torch.zeros((2 * total - 1, 4), dtype=torch.int64, device="cuda:0")
    one[:total] = dev_tensor(leaves)
    forest = torch.zeros((n_trees * (2 * m - 1), 4), dtype=torch.int64, device="cuda:0")
    forest[:total] = one[:total]
    torch.cuda.synchronize()
    ctx.merkle_2to1_dev(one.data_ptr(), total, stream)
    ctx.merkle_2to1_forest_dev(forest.data_ptr(), n_trees, m, stream)
    torch.cuda.synchronize()
    assert torch.equal(forest, one[:forest.shape[0]])                 # leaves + ten levels, level-major in both
    roots = to_numpy(forest[-n_trees:])
    for b in (0, 1, 777, n_trees - 1):
        assert np.array_equal(roots[b], c_oracle(name).merkle(leaves[b * m:(b + 1) * m], threads=0)[-1]), b
