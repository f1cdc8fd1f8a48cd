"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol, and its
host-only pieces (parameter generation, Montgomery conversion, validation) agree with the oracle.
No kernel runs here."""
import os
import re

import numpy as np
import pytest

import sponge_amd as S
from sponge_amd import _lib, synth
from oracle import kats as K
from oracle import poseidon_oracle as O

from helpers import golden, oracle_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="poseidon_mi355x.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 20
    lib = _lib.lib()
    for name in names:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.pmx_abi_version() == _lib.ABI_VERSION == 5
    # the benchmark diagnostics are a library of their own (include/poseidon_mi355x_diag.h): nothing of them in the shipped one
    diag = declared_functions("poseidon_mi355x_diag.h")
    assert sorted(_lib.DIAG_SIGNATURES) == diag and len(diag) == 3 and not set(diag) & set(names)
    for name in diag:
        assert not hasattr(lib, name), f"{name}: a benchmark diagnostic is exported by the shipped library"
    assert os.path.exists(_lib.DIAG_LIB_PATH), "make -C sponge_amd/csrc builds libposeidon_mi355x_diag.so"
    import ctypes
    d = ctypes.CDLL(_lib.DIAG_LIB_PATH)
    for name in diag:
        assert hasattr(d, name), name
    # the device-group test hooks have their own header and are NOT in the library that ships: not exported, not contained
    hooks = declared_functions("poseidon_mi355x_testing.h")
    assert sorted(_lib.TEST_HOOK_SIGNATURES) == hooks and not set(hooks) & set(names)
    for name in hooks:
        assert not hasattr(lib, name), f"{name}: a test hook is exported by the shipped library"
    blob = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"PMX_RCCL_LIBRARY", b"pmx_mgpu_test_", b"pmx_test_hooks_enabled", b"injected failure"):
        assert needle not in blob, needle


def test_the_test_library_is_the_shipped_abi_plus_the_hooks():
    """libposeidon_mi355x_test.so = the shipped objects with pmx_mgpu.cpp compiled -DPMX_TEST_HOOKS (csrc/Makefile): every symbol of
    the product header AND the hooks of include/poseidon_mi355x_testing.h; loaded side by side here (ctypes, no call into a device);
    the Rust binding declares none of the hooks."""
    import ctypes
    assert os.path.exists(_lib.TEST_LIB_PATH), "make -C sponge_amd/csrc builds both libraries"
    t = ctypes.CDLL(_lib.TEST_LIB_PATH)
    for name in declared_functions() + declared_functions("poseidon_mi355x_testing.h"):
        assert hasattr(t, name), name
    t.pmx_abi_version.restype = ctypes.c_int
    assert t.pmx_abi_version() == _lib.ABI_VERSION
    assert t.pmx_test_hooks_enabled() == 1
    ffi = open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read()
    assert "pmx_mgpu_test" not in ffi and "pmx_test_hooks" not in ffi


def test_rust_binding_source_declares_the_whole_header():
    """bindings/rust cannot be compiled in this image (no rustc): at least keep its extern block complete."""
    ffi = open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read()
    missing = [n for n in declared_functions() if f"fn {n}(" not in ffi]
    assert not missing, missing


def test_integration_md_shows_the_binding_files_verbatim():
    """INTEGRATION.md embeds bindings/rust/{build.rs, src/ffi.rs, src/mod.rs}: the document and the files must not drift."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for rel in ("build.rs", os.path.join("src", "ffi.rs"), os.path.join("src", "mod.rs"), os.path.join("src", "tests.rs")):
        text = open(os.path.join(ROOT, "bindings", "rust", rel)).read().rstrip("\n")
        assert "```rust\n" + text + "\n```" in doc, rel


def test_rust_binding_tests_restate_the_reference_kat_and_a_differential_test():
    """bindings/rust/src/tests.rs cannot run here; keep what it must contain from drifting: the three KAT outputs of
    src/poseidon/mod.rs:376-399 (data, the same numbers oracle/kats.py pins), a differential test against the crate's own
    PoseidonSponge, and the `mod tests;` hook in mod.rs."""
    text = open(os.path.join(ROOT, "bindings", "rust", "src", "tests.rs")).read()
    for value in K.SPONGE_CONSISTENCY_OUTPUT:
        assert f'MontFp!("{value}")' in text, value
    assert "PoseidonSponge::<Fr>::new" in text and "Mi355xPoseidonSponge::<Fr>::new" in text and "fn differential_" in text
    assert "mod tests;" in open(os.path.join(ROOT, "bindings", "rust", "src", "mod.rs")).read()
    assert "mi355x" in open(os.path.join(ROOT, "bindings", "rust", "Cargo.fragment.toml")).read()


def test_no_device_means_loud_failure_not_fallback():
    if _lib.lib().pmx_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = S.poseidon_config_from_lfsr(S.BLS12_381_FR, 2, 5, 8, 31)
    with pytest.raises(S.PmxError) as ei:
        cfg.context(0)
    assert ei.value.code == _lib.PMX_ERR_HIP
    assert "no CPU fallback" in str(ei.value)


@pytest.mark.parametrize("field,ofield", [(S.BLS12_381_FR, O.BLS12_381_FR), (S.BN254_FR, O.BN254_FR)])
def test_montgomery_constants_and_conversion(field, ofield):
    inv, r, r2 = field.mont_constants()
    mc = O.mont_constants(ofield)
    assert inv == mc["inv"]
    assert [int(x) for x in r] == O.to_limbs(mc["r"])
    assert [int(x) for x in r2] == O.to_limbs(mc["r2"])
    vals = [0, 1, 2, ofield - 1, ofield // 3, 1 << 200]
    m = field.from_ints(vals)
    assert [O.from_limbs([int(x) for x in row]) for row in m] == [O.to_mont(v, ofield) for v in vals]
    assert field.to_ints(m) == vals


def test_unreduced_element_is_rejected():
    f = S.BLS12_381_FR
    bad = np.array([[0xFFFFFFFFFFFFFFFF] * 4], dtype=np.uint64)
    with pytest.raises(S.PmxError):
        f.to_ints(bad)


def test_lfsr_reference_kats_through_the_library():
    # src/poseidon/grain_lfsr.rs:197-213 and src/poseidon/traits.rs:163-358, via pmx_find_poseidon_ark_and_mds
    f = S.BLS12_381_FR
    for (rate, weights), (ark00, mds00) in K.DEFAULT_PARAMS_BLS12_381.items():
        cfg = S.get_default_poseidon_parameters(f, rate, weights)
        assert cfg.capacity == 1 and cfg.rate == rate
        assert f.to_ints(cfg.ark[0, 0:1]) == [ark00]
        assert f.to_ints(cfg.mds[0, 0:1]) == [mds00]
    cfg = S.get_default_poseidon_parameters(f, 2, False)
    assert f.to_ints(cfg.ark[0, 0:2]) == K.GRAIN_LFSR_255_3_8_31
    assert (cfg.alpha, cfg.full_rounds, cfg.partial_rounds) == (17, 8, 31)
    assert S.get_default_poseidon_parameters(f, 9, False) is None
    assert S.get_default_poseidon_parameters(S.BN254_FR, 2, False) is None


@pytest.mark.parametrize("name", sorted(golden("config_pins.json")))
def test_generated_constants_equal_oracle(name):
    pin = golden("config_pins.json")[name]
    f = S.FIELDS[pin["field"]]
    cfg = S.poseidon_config_from_lfsr(f, pin["rate"], pin["alpha"], pin["full_rounds"], pin["partial_rounds"])
    ocfg = oracle_config(name)
    assert f.to_ints(cfg.ark.reshape(-1, 4)) == [v for row in ocfg.ark for v in row]
    assert f.to_ints(cfg.mds.reshape(-1, 4)) == [v for row in ocfg.mds for v in row]
    assert f.to_ints(cfg.ark[0, 0:1]) == [int(pin["ark_first"], 16)]
    assert f.to_ints(cfg.mds[-1, -1:]) == [int(pin["mds_last"], 16)]


def test_parameter_errors():
    f = S.BLS12_381_FR
    with pytest.raises(S.PmxError):   # prime_bits must equal the modulus bit size (grain_lfsr.rs:112)
        S.find_poseidon_ark_and_mds(f, 254, 2, 8, 31, 0)
    with pytest.raises(AssertionError):   # PoseidonConfig::new asserts, mod.rs:196-203
        S.PoseidonConfig(f, 8, 31, 5, np.zeros((3, 3, 4), np.uint64), np.zeros((38, 3, 4), np.uint64), 2, 1)
    with pytest.raises(AssertionError):
        S.PoseidonConfig(f, 8, 31, 5, np.zeros((3, 2, 4), np.uint64), np.zeros((39, 3, 4), np.uint64), 2, 1)


def test_synthetic_batches_are_reduced_and_shardable():
    for f in (S.BLS12_381_FR, S.BN254_FR):
        a = synth.random_elements(f, 1000, seed=0x5EED0001)
        vals = [O.from_limbs([int(x) for x in row]) for row in a]
        assert all(v < f.modulus for v in vals)
        assert len(set(vals)) == 1000
        b = synth.random_elements(f, 400, seed=0x5EED0001, offset=600)
        assert np.array_equal(a[600:], b)
    # splitmix64 known answers (seed 0 stream: 0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4)
    out = synth.splitmix64(np.array([0, 1], dtype=np.uint64))
    assert [int(x) for x in out] == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4]


def _try_create(cfg_kwargs):
    """pmx_ctx_create validates the config before it looks for a device, so the reference's
    PoseidonConfig::new failures (and this build's limits) are observable without a GPU."""
    import ctypes
    f = cfg_kwargs.pop("field", S.BLS12_381_FR)
    t = cfg_kwargs["rate"] + cfg_kwargs["capacity"]
    rounds = cfg_kwargs["full_rounds"] + cfg_kwargs["partial_rounds"]
    ark = cfg_kwargs.pop("ark", np.zeros((rounds, t, 4), dtype=np.uint64))
    mds = cfg_kwargs.pop("mds", np.zeros((t, t, 4), dtype=np.uint64))
    modulus = cfg_kwargs.pop("modulus", f.modulus)
    c = _lib.PmxConfig()
    c.full_rounds, c.partial_rounds = cfg_kwargs["full_rounds"], cfg_kwargs["partial_rounds"]
    c.alpha, c.rate, c.capacity = cfg_kwargs.get("alpha", 5), cfg_kwargs["rate"], cfg_kwargs["capacity"]
    for i in range(4):
        c.modulus[i] = (modulus >> (64 * i)) & (2**64 - 1)
    c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
    h = ctypes.c_void_p()
    rc = _lib.lib().pmx_ctx_create(ctypes.byref(c), 0, ctypes.byref(h))
    msg = _lib.lib().pmx_last_error().decode()
    if rc == 0:
        _lib.lib().pmx_ctx_destroy(h)
    return rc, msg


@pytest.mark.parametrize("kwargs,code,needle", [
    (dict(full_rounds=8, partial_rounds=31, rate=0, capacity=1), _lib.PMX_ERR_CONFIG, "rate"),
    (dict(full_rounds=8, partial_rounds=31, rate=16, capacity=1), _lib.PMX_ERR_UNSUPPORTED, "width"),
    (dict(full_rounds=0, partial_rounds=0, rate=2, capacity=1), _lib.PMX_ERR_CONFIG, "round"),
    (dict(full_rounds=8, partial_rounds=31, rate=2, capacity=1, modulus=(1 << 256) - 189), _lib.PMX_ERR_UNSUPPORTED, "2^255"),
    (dict(full_rounds=8, partial_rounds=31, rate=2, capacity=1, modulus=(1 << 254)), _lib.PMX_ERR_CONFIG, "odd"),
    (dict(full_rounds=8, partial_rounds=31, rate=2, capacity=1, modulus=(1 << 61) - 1), _lib.PMX_ERR_UNSUPPORTED, "225"),
])
def test_config_validation_without_a_device(kwargs, code, needle):
    rc, msg = _try_create(dict(kwargs))
    assert rc == code and needle in msg, (rc, msg)


def test_unreduced_constants_are_rejected():
    bad = np.zeros((39, 3, 4), dtype=np.uint64)
    bad[5, 1] = np.uint64(0xFFFFFFFFFFFFFFFF)          # >= p
    rc, msg = _try_create(dict(full_rounds=8, partial_rounds=31, rate=2, capacity=1, ark=bad))
    assert rc == _lib.PMX_ERR_CONFIG and "ark constant 16 is not reduced" in msg
    rc, msg = _try_create(dict(full_rounds=8, partial_rounds=31, rate=2, capacity=1))
    if _lib.lib().pmx_device_count() == 0:            # a valid config then fails only for want of a device
        assert rc == _lib.PMX_ERR_HIP


def test_merkle_paths_gather_is_host_only_and_matches_the_oracle_tree():
    """pmx_merkle_paths over a node array built by the oracle's C restatement (no device involved): every sibling is
    the node the index arithmetic says, recomputing the path with the oracle's 2-to-1 hash reaches the root."""
    import ctypes
    from oracle import cref
    cr = cref.CRef(oracle_config("bls_t3_a5_8_31"))
    m, depth = 64, 6
    leaves = synth.random_elements(S.BLS12_381_FR, m, seed=99)
    nodes = cr.merkle(leaves, threads=1)
    idx = np.array([0, 1, 37, 63], dtype=np.uint64)
    paths = np.zeros((4, depth, 4), dtype=np.uint64)
    lib = _lib.lib()
    _lib.check(lib.pmx_merkle_paths(ctypes.c_void_p(nodes.ctypes.data), m, ctypes.c_void_p(idx.ctypes.data), 4,
                                    ctypes.c_void_p(paths.ctypes.data)))
    for row, leaf_index in zip(paths, idx):
        cur, i = leaves[int(leaf_index)], int(leaf_index)
        for level in range(depth):
            pair = np.stack([row[level], cur] if i & 1 else [cur, row[level]]).reshape(1, 2, 4)
            cur = cr.hash_batch(pair, 2, 1, threads=1).reshape(4)
            i >>= 1
        assert np.array_equal(cur, nodes[-1])
    bad = np.array([64], dtype=np.uint64)
    assert lib.pmx_merkle_paths(ctypes.c_void_p(nodes.ctypes.data), m, ctypes.c_void_p(bad.ctypes.data), 1,
                                ctypes.c_void_p(paths.ctypes.data)) == _lib.PMX_ERR_ARG
    assert lib.pmx_merkle_paths(ctypes.c_void_p(nodes.ctypes.data), 48, ctypes.c_void_p(idx.ctypes.data), 1,
                                ctypes.c_void_p(paths.ctypes.data)) == _lib.PMX_ERR_ARG


def test_host_allocation_failure_becomes_a_status_code():
    """include/poseidon_mi355x.h: "nothing throws or aborts across the boundary".  A child process loads the library,
    then caps its address space just above what it already uses and asks for a context whose tables need a few MB
    (4000 rounds of width 16): the std::bad_alloc inside prepare() must come back as PMX_ERR_HOST with a message, not as
    std::terminate (which would kill the child with SIGABRT)."""
    import subprocess
    import sys
    code = r'''
import ctypes, resource, sys
import numpy as np
sys.path.insert(0, %r)
from sponge_amd import _lib
import sponge_amd as S
lib = _lib.lib()
f = S.BLS12_381_FR
rounds, t = 4000, 16
ark = np.zeros((rounds, t, 4), dtype=np.uint64)
mds = np.zeros((t, t, 4), dtype=np.uint64)
c = _lib.PmxConfig()
c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity = 8, rounds - 8, 5, t - 1, 1
for i in range(4):
    c.modulus[i] = (f.modulus >> (64 * i)) & (2**64 - 1)
c.ark, c.mds = ark.ctypes.data, mds.ctypes.data
h = ctypes.c_void_p()
vm = [int(l.split()[1]) * 1024 for l in open("/proc/self/status") if l.startswith("VmSize")][0]
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
resource.setrlimit(resource.RLIMIT_AS, (vm + (1 << 20), hard))
rc = lib.pmx_ctx_create(ctypes.byref(c), 0, ctypes.byref(h))
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
msg = lib.pmx_last_error().decode()
print(rc, "|", msg)
sys.exit(0 if (rc == _lib.PMX_ERR_HOST and "out of host memory" in msg) else 1)
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])
