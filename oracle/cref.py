"""ctypes loader for oracle/libposeidon_oracle.so (the C restatement)  --  TEST INFRASTRUCTURE ONLY.

Used by tests/ (bulk comparator), __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

from . import poseidon_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libposeidon_oracle.so")


class _Config(ctypes.Structure):
    _fields_ = [
        ("t", ctypes.c_uint32), ("rate", ctypes.c_uint32), ("capacity", ctypes.c_uint32),
        ("full_rounds", ctypes.c_uint32), ("partial_rounds", ctypes.c_uint32), ("pad_", ctypes.c_uint32),
        ("alpha", ctypes.c_uint64), ("modulus", ctypes.c_uint64 * 4), ("inv", ctypes.c_uint64),
        ("ark", ctypes.c_void_p), ("mds", ctypes.c_void_p),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "poseidon_ref.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.pref_max_threads.restype = ctypes.c_int
    return _lib


_native = None


def native_lib():
    """The same C restatement compiled on THIS machine with -march=native (BASELINE.md section 2 quotes the CPU path with
    -march=native): bench.py's cpu_baseline uses it when gcc is there, so that the CPU figure is the host's own ceiling and
    not that of the portable build (x86-64-v3: no ADX) which has to travel between machines.  Built into a private
    temporary directory, never into the repo; None when it cannot be built.  The checker proper (tests, smoke, bench.py's
    verification) stays on the portable build."""
    global _native
    if _native is None:
        import shutil
        import tempfile
        _native = False
        cc = shutil.which("gcc") or shutil.which("cc")
        if cc:
            d = tempfile.mkdtemp(prefix="pmx_oracle_native_")
            out = os.path.join(d, "libposeidon_oracle_native.so")
            cmd = [cc, "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=c11", "-shared", "-o", out, os.path.join(HERE, "poseidon_ref.c")]
            try:
                subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
                h = ctypes.CDLL(out)
                h.pref_max_threads.restype = ctypes.c_int
                _native = h
            except Exception:
                _native = False
    return _native or None


def elems_to_limbs(vals, p: int, mont: bool = True) -> np.ndarray:
    """canonical ints (any nesting flattened by the caller) -> [n][4] u64 (Montgomery by default)."""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        if mont:
            v = O.to_mont(v, p)
        out[i] = O.to_limbs(v)
    return out


def limbs_to_elems(arr: np.ndarray, p: int, mont: bool = True):
    arr = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    out = []
    for row in arr:
        v = O.from_limbs([int(x) for x in row])
        out.append(O.from_mont(v, p) if mont else v)
    return out


class CRef:
    """One PoseidonConfig loaded into the C restatement."""

    def __init__(self, cfg: O.PoseidonConfig, handle=None):
        self.cfg = cfg
        self._h = handle            # None: the portable build (lib()); bench.py's cpu_baseline passes native_lib()
        p = cfg.p
        self._ark = np.ascontiguousarray(
            elems_to_limbs([v for row in cfg.ark for v in row], p).reshape(-1))
        self._mds = np.ascontiguousarray(
            elems_to_limbs([v for row in cfg.mds for v in row], p).reshape(-1))
        c = _Config()
        c.t, c.rate, c.capacity = cfg.t, cfg.rate, cfg.capacity
        c.full_rounds, c.partial_rounds, c.alpha = cfg.full_rounds, cfg.partial_rounds, cfg.alpha
        for i, l in enumerate(O.to_limbs(p)):
            c.modulus[i] = l
        c.inv = O.mont_constants(p)["inv"]
        c.ark = self._ark.ctypes.data
        c.mds = self._mds.ctypes.data
        self._c = c

    @staticmethod
    def _ptr(a: np.ndarray):
        assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
        return ctypes.c_void_p(a.ctypes.data)

    def permute_batch(self, states: np.ndarray, threads: int = 1) -> np.ndarray:
        """states [n][t][4] Montgomery limbs -> permuted copy.  threads=0: every usable CPU."""
        threads = threads or max_threads()
        out = np.ascontiguousarray(states, dtype=np.uint64).copy()
        n = out.size // (self.cfg.t * 4)
        rc = (self._h or lib()).pref_permute_batch(ctypes.byref(self._c), self._ptr(out), ctypes.c_size_t(n),
                                                   ctypes.c_int(threads))
        assert rc == 0
        return out

    def hash_batch(self, msgs: np.ndarray, L: int, k: int, threads: int = 1) -> np.ndarray:
        threads = threads or max_threads()
        msgs = np.ascontiguousarray(msgs, dtype=np.uint64)
        n = msgs.size // (L * 4) if L else msgs.shape[0]
        out = np.zeros((n, k, 4), dtype=np.uint64)
        rc = lib().pref_hash_batch(ctypes.byref(self._c), self._ptr(msgs), ctypes.c_size_t(L),
                                   self._ptr(out), ctypes.c_size_t(k), ctypes.c_size_t(n),
                                   ctypes.c_int(threads))
        assert rc == 0
        return out

    def merkle(self, leaves: np.ndarray, threads: int = 1) -> np.ndarray:
        """leaves [m][4] -> nodes [2m-1][4] (leaves, then each level, root last)."""
        threads = threads or max_threads()
        leaves = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, 4)
        m = leaves.shape[0]
        nodes = np.zeros((2 * m - 1, 4), dtype=np.uint64)
        nodes[:m] = leaves
        rc = lib().pref_merkle_2to1(ctypes.byref(self._c), self._ptr(nodes), ctypes.c_size_t(m),
                                    ctypes.c_int(threads))
        assert rc == 0
        return nodes

    def sponge_absorb(self, state: np.ndarray, mode: int, index: int, elems: np.ndarray):
        state = np.ascontiguousarray(state, dtype=np.uint64).copy()
        elems = np.ascontiguousarray(elems, dtype=np.uint64).reshape(-1, 4)
        m, i = ctypes.c_uint32(mode), ctypes.c_uint32(index)
        lib().pref_sponge_absorb(ctypes.byref(self._c), self._ptr(state), ctypes.byref(m),
                                 ctypes.byref(i), self._ptr(elems), ctypes.c_size_t(elems.shape[0]))
        return state, m.value, i.value

    def sponge_squeeze(self, state: np.ndarray, mode: int, index: int, n: int):
        state = np.ascontiguousarray(state, dtype=np.uint64).copy()
        out = np.zeros((n, 4), dtype=np.uint64)
        m, i = ctypes.c_uint32(mode), ctypes.c_uint32(index)
        lib().pref_sponge_squeeze(ctypes.byref(self._c), self._ptr(state), ctypes.byref(m),
                                  ctypes.byref(i), self._ptr(out), ctypes.c_size_t(n))
        return state, m.value, i.value, out


def usable_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (cpu.max), if any."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.999)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.999)))
            break
        except Exception:
            continue
    return n


def max_threads() -> int:
    """Threads worth using: OpenMP's maximum capped by the CPUs this process may actually use."""
    return max(1, min(int(lib().pref_max_threads()), usable_cpus()))
