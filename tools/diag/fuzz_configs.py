#!/usr/bin/env python3
"""Seeded random Poseidon configurations on the GPU against the C port (oracle/), aimed at ENGINES: every exponent class (0, 1, small,
the usual, 64-bit), odd and zero round counts, every rate / capacity split of widths 2 ... 12, both benchmarked fields and (every fourth
config) a random prime of 225 ... 255 bits - every other one of those with top byte 127, whose larger residues the int8 tables store as
Y - p; per config
whole permutations at several batch sizes, the fixed-shape hash, a small tree and (every other config) the duplex driver on sponges in
mixed modes; at t = 3 one LARGE call per config (32769 ... 2^18 + units: the other side of the quad kernels' threshold in
pmx_device.hip), rotating through permute / hash / compress / absorb + squeeze.

pmx_ctx_engine_info is asked before every call, and the run keeps the matrix  engine family x operation -> calls checked.  With --matrix
(what tests/test_gpu_fuzz.py passes) the matrix is printed and every cell of REQUIRED must have been hit, or the run fails: a family
that the random choices never reached is a family nothing tested.  Prints one line per failing case and a summary; exit code 1 if
anything differs or a required cell is empty.      usage: tools/diag/fuzz_configs.py [n_configs] [seed] [--matrix]"""
import collections
import ctypes
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sponge_amd as S
from sponge_amd import _lib, synth
from oracle import cref, poseidon_oracle as O

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 120
SEED = int(args[1]) if len(args) > 1 else 2024
WANT_MATRIX = "--matrix" in sys.argv
rng = random.Random(SEED)
FIELDS = {"bls12_381_fr": (O.BLS12_381_FR, 255), "bn254_fr": (O.BN254_FR, 254)}
OPS = {_lib.OP_PERMUTE: "permute", _lib.OP_HASH: "hash", _lib.OP_COMPRESS: "compress", _lib.OP_ABSORB: "absorb", _lib.OP_SQUEEZE: "squeeze"}
# engine family (pmx_ctx_engine_info's name, reduced: see family()) x operation: every cell must be reached when --matrix is given
FAMILIES = ["QuadEngine", "HybridEngine mfma", "HybridEngine mfma x passes", "LdsEngine"]
REQUIRED = {(fam, op) for fam in FAMILIES for op in OPS.values()}
# cells that do not exist: the pass form is the absorb / squeeze driver, and the only form of it on the window engines
REQUIRED -= {("HybridEngine mfma x passes", op) for op in ("permute", "hash", "compress")}
REQUIRED -= {("HybridEngine mfma", op) for op in ("absorb", "squeeze")}
OPTIONAL = set()
matrix = collections.Counter()


def family(info):
    name = info.engine.decode()
    head = name.split("<")[0]
    if head == "HybridEngine":
        head += " mfma" if "mfma" in name else " ?"
    if info.launches > 1 or name.endswith("x passes"):
        head += " x passes"
    return head


def engine(ctx, op, n, length=0):
    info = _lib.PmxEngineInfo()
    _lib.check(_lib.lib().pmx_ctx_engine_info(ctx._h, op, n, length, ctypes.byref(info)))
    matrix[(family(info), OPS[op])] += 1
    return info.engine.decode()


def _is_prime(n):
    for q in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        if n % q == 0:
            return n == q
    d, r = n - 1, 0
    while d % 2 == 0:
        d //= 2
        r += 1
    for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(r - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def random_prime(bits, top_byte=None):
    """a prime of exactly `bits` bits (the library takes 225 ... 255: pmx_prepare.hpp); top_byte: its bits 248 ... 255"""
    while True:
        c = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        if top_byte is not None:
            c = (c & ((1 << 248) - 1)) | (top_byte << 248)
        if _is_prime(c):
            return c


ALPHAS = [0, 1, 1, 2, 3, 4, 5, 5, 5, 6, 7, 11, 17, 17, 257, 65537, (1 << 32) + 1, (1 << 63) + 1, (1 << 64) - 1]
# The first configurations of every run are DIRECTED: the rare corners of the dispatch that random draws reach once in a thousand configs
# (field: a name of FIELDS or "p127" = a random 255-bit prime with top byte 127; large = (operation, units) of the one large call at t = 3).
# The constants, states and messages stay seeded-random; the rest of the run is random in everything.
DIRECTED = [
    dict(field="p127", t=3, capacity=1, alpha=5, rf=8, rp=31, large=("permute", 32769)),            # residues stored as Y - p (pmx_prepare.hpp): t = 3 window engine
    dict(field="p127", t=3, capacity=1, alpha=17, rf=8, rp=31, large=("hash", 40001)),
    dict(field="p127", t=3, capacity=1, alpha=5, rf=8, rp=31, large=("tree", 1 << 17)),
    dict(field="p127", t=3, capacity=1, alpha=5, rf=8, rp=31, large=("sponges", 32769 + 77)),
    dict(field="p127", t=3, capacity=1, alpha=3, rf=8, rp=31, large=("permute", 32769 + 130)),      # ... the generic S-box
    dict(field="p127", t=3, capacity=1, alpha=5, rf=8, rp=31, large=("hash", (1 << 17) + 1)),
    dict(field="p127", t=3, capacity=1, alpha=7, rf=8, rp=31, large=("tree", 1 << 17)),
    dict(field="p127", t=3, capacity=1, alpha=257, rf=8, rp=13, large=("sponges", 32769 + 200)),
    dict(field="bls12_381_fr", t=3, capacity=1, alpha=5, rf=8, rp=31, large=("sponges", 32769 + 5)),  # the window engine at t = 3, as passes
    dict(field="bls12_381_fr", t=3, capacity=1, alpha=17, rf=8, rp=31, large=("tree", 1 << 17)),
    dict(field="bn254_fr", t=3, capacity=1, alpha=5, rf=8, rp=57, large=("hash", 32769)),
    dict(field="bls12_381_fr", t=3, capacity=1, alpha=5, rf=8, rp=0, large=("permute", 32769)),       # no partial section: the dense schedule (run-time-width engine)
    dict(field="bls12_381_fr", t=3, capacity=0, alpha=5, rf=8, rp=31, large=("hash", 20000)),         # a split the quad kernels do not take
    dict(field="p127", t=5, capacity=1, alpha=5, rf=8, rp=56, large=None),                            # the wide window engines on such a modulus
    dict(field="p127", t=9, capacity=1, alpha=17, rf=8, rp=57, large=None),
    dict(field="p127", t=7, capacity=2, alpha=5, rf=7, rp=22, large=None),
    dict(field="bn254_fr", t=9, capacity=1, alpha=5, rf=8, rp=57, large=None),                        # BASELINE configs[2]'s shape
    dict(field="bls12_381_fr", t=2, capacity=1, alpha=5, rf=8, rp=56, large=None),                    # the run-time-width engine
    dict(field="bn254_fr", t=12, capacity=1, alpha=5, rf=8, rp=57, large=None),
    dict(field="bls12_381_fr", t=6, capacity=1, alpha=5, rf=8, rp=0, large=None),
]
bad = 0
t0 = time.time()
large_turn = 0


def fail(msg):
    global bad
    bad += 1
    print(msg + "   [seed %d]" % SEED, flush=True)


def check_sponges(what, f, cfg, cr, n, t, rate, script, seed, sample=None):
    """the duplex driver on n sponges in random modes and positions; `sample`: the sponges checked one by one against the C port (all if None)"""
    nrng = np.random.default_rng(seed)
    batch = S.BatchPoseidonSponge.new(cfg, n)
    batch.state = synth.random_elements(f, n * t, seed=seed + 11).reshape(n, t, 4)
    batch.mode_tag = nrng.integers(0, 2, n).astype(np.uint32)
    batch.mode_index = nrng.integers(0, rate + 1, n).astype(np.uint32)
    idx = list(range(n)) if sample is None else sample
    ref = {i: (batch.state[i].copy(), int(batch.mode_tag[i]), int(batch.mode_index[i])) for i in idx}
    ctx = cfg.context()
    for (op, length) in script:
        ok = True
        if op == "absorb":
            eng = engine(ctx, _lib.OP_ABSORB, n, length)
            elems = synth.random_elements(f, n * length, seed=length + seed).reshape(n, length, 4)
            batch.absorb(elems)
            ref = {j: cr.sponge_absorb(st, m, i, elems[j]) for j, (st, m, i) in ref.items()}
        else:
            eng = engine(ctx, _lib.OP_SQUEEZE, n, length)
            out = batch.squeeze_native_field_elements(length)
            nxt = {}
            for j, (st, m, i) in ref.items():
                s2, m2, i2, o = cr.sponge_squeeze(st, m, i, length)
                ok = ok and np.array_equal(out[j], o)
                nxt[j] = (s2, m2, i2)
            ref = nxt
        for j, (st, m, i) in ref.items():
            ok = ok and np.array_equal(batch.state[j], st) and (batch.mode_tag[j], batch.mode_index[j]) == (m, i)
        if not ok:
            fail("SPONGE %s(%d) differs: %s n=%d engine=%s" % (op, length, what, n, eng))


for k in range(N):
    d = DIRECTED[k] if k < len(DIRECTED) else None
    if d is not None and d["field"] == "p127":
        bits, p = 255, random_prime(255, top_byte=127)
        fname = "prime%d_%x" % (bits, p >> (bits - 16))
        f = S.Field(fname, p)
    elif d is not None:
        fname = d["field"]
        p, bits = FIELDS[fname]
        f = S.FIELDS[fname]
    elif k % 4 == 3:                     # every fourth config over a random prime: any size the library takes, any top byte
        if (k // 4) % 2 == 0:            # ... every other one of them a modulus WITHOUT int8 tables (top byte 127: pmx_mfma.hpp)
            bits, p = 255, random_prime(255, top_byte=127)
        else:
            bits = rng.choice([225, 226, 233, 240, 247, 248, 249, 253, 254, 255])
            p = random_prime(bits)
        fname = "prime%d_%x" % (bits, p >> (bits - 16))
        f = S.Field(fname, p)
    else:
        fname = rng.choice(list(FIELDS))
        p, bits = FIELDS[fname]
        f = S.FIELDS[fname]
    t = rng.choice([2, 3, 3, 3, 4, 5, 6, 7, 8, 9, 9, 10, 12])
    capacity = rng.choice([1, 1, 1, 0, 2, 3])
    capacity = min(capacity, t - 1)
    rate = t - capacity
    alpha = rng.choice(ALPHAS)
    rf = rng.choice([0, 1, 2, 3, 4, 6, 7, 8, 8, 8, 10])
    rp = rng.choice([0, 1, 2, 3, 5, 6, 7, 13, 22, 31, 56, 57, 60, 66, 67, 70])
    if rf + rp == 0:
        rf = 2
    if d is not None:
        t, capacity, alpha, rf, rp = d["t"], d["capacity"], d["alpha"], d["rf"], d["rp"]
        rate = t - capacity
    base = S.poseidon_config_from_lfsr(f, t - 1, alpha, rf, rp)
    cfg = S.PoseidonConfig(f, rf, rp, alpha, base.mds, base.ark, rate, capacity)
    ob = O.make_config(p, bits, t - 1, alpha, rf, rp)
    cr = cref.CRef(O.PoseidonConfig(ob.p, rf, rp, alpha, ob.ark, ob.mds, rate, capacity))
    ctx = cfg.context()
    what = "%s t=%d rate=%d cap=%d alpha=%d rf=%d rp=%d" % (fname, t, rate, capacity, alpha, rf, rp)

    def permute_case(n):
        eng = engine(ctx, _lib.OP_PERMUTE, n)
        states = synth.random_elements(f, n * t, seed=k * 7 + n).reshape(n, t, 4)
        states[0] = 0
        if n > 1:
            states[1] = f.from_ints([p - 1] * t)
        if n > 2:
            states[2, 0] = 0
        got, want = ctx.permute_batch(states), cr.permute_batch(states, threads=0)
        if not np.array_equal(got, want):
            fail("PERMUTE differs: %s n=%d engine=%s (%d states)" % (what, n, eng, int((got != want).any(axis=(1, 2)).sum())))

    def hash_case(n, L, ko):
        eng = engine(ctx, _lib.OP_HASH, n, L)
        msgs = synth.random_elements(f, max(n * L, 1), seed=k).reshape(n, L, 4) if L else np.zeros((n, 0, 4), dtype=np.uint64)
        try:
            got, want = ctx.hash_batch(msgs, L, ko, n), cr.hash_batch(msgs, L, ko, threads=0)
            if not np.array_equal(got, want):
                fail("HASH differs: %s L=%d k=%d n=%d engine=%s" % (what, L, ko, n, eng))
        except Exception as e:
            fail("HASH raised: %s L=%d k=%d n=%d engine=%s: %s" % (what, L, ko, n, eng, e))

    def tree_case(n_leaves):
        width = n_leaves // 2
        while width >= 1:                # (the level-by-level launches of pmx_merkle_2to1: one engine question per level width)
            eng = engine(ctx, _lib.OP_COMPRESS, width)
            width //= 2
        leaves = synth.random_elements(f, n_leaves, seed=k + 5)
        nodes, _ = ctx.merkle_2to1(leaves)
        if not np.array_equal(nodes, cr.merkle(leaves, threads=0)):
            fail("MERKLE differs: %s leaves=%d" % (what, n_leaves))

    for n in (1, 67, 300):
        permute_case(n)
    L, ko = rng.randint(0, 2 * t + 1), rng.randint(0, t + 2)
    if L + ko > 0:
        hash_case(150, L, ko)
    if rate >= 2:
        tree_case(128)
    if (k % 3 != 1 or d is not None) and rate >= 1:
        r = rate
        check_sponges(what, f, cfg, cr, 90, t, rate,
                      [("absorb", rng.randint(1, 2 * r + 1)), ("squeeze", rng.randint(1, 2 * r + 1)), ("squeeze", 1), ("absorb", 1)], k)
    if t == 3:
        # one LARGE call: the far side of the t = 3 threshold (32769: quad kernels -> window engine)
        turn, large_turn = large_turn % 5, large_turn + 1
        pick = {0: "permute", 1: "hash", 2: "tree"}.get(turn, "sponges")
        big = None
        if d is not None:
            pick, big = d["large"] if d["large"] else (None, None)
        if pick == "permute":
            permute_case(big or rng.choice([32769, 32769 + 64 * rng.randint(1, 50) + 3, (1 << 17) + 3]))
        elif pick == "hash":
            hash_case(big or rng.choice([32769, 40001, (1 << 17) + 1]), rng.randint(1, 5), rng.randint(1, 3))
        elif pick == "tree" and rate >= 2:
            tree_case(big or rng.choice([1 << 17, 1 << 19]))      # levels of 2^16 (and 2^18) compressions on the large side, the rest below
        elif pick == "sponges" and rate >= 1:
            n = big or 32769 + 64 * rng.randint(0, 20) + rng.randint(0, 63)
            srng = random.Random(k)
            sample = sorted(set(list(range(64)) + list(range(n - 64, n)) + [srng.randrange(n) for _ in range(300)]))
            check_sponges(what, f, cfg, cr, n, t, rate, [("absorb", rng.randint(1, 2 * rate + 1)), ("squeeze", rng.randint(1, 2 * rate + 1))], k + 1,
                          sample=sample)
    ctx.close() if hasattr(ctx, "close") else None

missing = []
if WANT_MATRIX:
    fams = sorted({fam for fam, _ in matrix} | set(FAMILIES))
    print("engine family x operation -> calls checked against the C port (%d configurations, seed %d)" % (N, SEED))
    print("  %-30s" % "" + "".join("%10s" % op for op in OPS.values()))
    for fam in fams:
        print("  %-30s" % fam + "".join("%10s" % (matrix[(fam, op)] if (fam, op) in REQUIRED or matrix[(fam, op)] else "-") for op in OPS.values()))
    missing = sorted(c for c in REQUIRED - OPTIONAL if matrix[c] == 0)
    unknown = sorted(c for c in matrix if c[0] not in FAMILIES)
    if unknown:
        print("engine families this tool does not know (add them to FAMILIES): %s" % unknown)
        missing += unknown
    if missing:
        print("EMPTY required cells: %s" % missing)
print("fuzz_configs: %d configurations, %d failing cases, %d empty cells, %.0f s" % (N, bad, len(missing), time.time() - t0))
sys.exit(1 if bad or missing else 0)
