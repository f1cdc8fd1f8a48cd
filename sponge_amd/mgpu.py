"""Device groups over the C ABI (pmx_mgpu_*): the batch sharded over the GPUs of one node, RCCL for the final
gather.  The reference has no distributed code - sponge states are independent (src/poseidon/mod.rs:62-183 has no
cross-state data flow) - so shards are contiguous and the permutation itself runs with no collective.

Two ways to form a group, as in the header:
  DeviceGroup.single_process(cfg, n_devices)          all GPUs of this process (ncclCommInitAll)
  DeviceGroup.one_rank(cfg, device, rank, world, id)  one rank of a multi-process job (ncclCommInitRank);
                                                      rank 0 makes `id` with unique_id() and the launcher
                                                      (torch.distributed in bench.py) carries it to the others
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .poseidon import Context, PoseidonConfig, c_config


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """(start, count) of rank's contiguous shard of n units (pmx_shard_bounds; host arithmetic, no device)."""
    start, count = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.lib().pmx_shard_bounds(n, world, rank, ctypes.byref(start), ctypes.byref(count)))
    return start.value, count.value


def unique_id() -> bytes:
    buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES)()
    _lib.check(_lib.lib().pmx_mgpu_unique_id(buf))
    return bytes(buf)


def _ptr_array(ptrs: Sequence[int]):
    return (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(int(p)) for p in ptrs])


class DeviceGroup:
    def __init__(self, handle, cfg: PoseidonConfig):
        self._h = handle
        self.cfg = cfg
        info = self.info()
        self.world, self.n_local, self.first_rank = info["world"], info["n_local"], info["first_rank"]
        self.devices = info["devices"]

    @classmethod
    def single_process(cls, cfg: PoseidonConfig, n_devices: Optional[int] = None,
                       devices: Optional[Sequence[int]] = None) -> "DeviceGroup":
        if devices is not None:
            n_devices = len(devices)
        if n_devices is None:
            n_devices = _lib.lib().pmx_device_count()
        arr = (ctypes.c_int * n_devices)(*devices) if devices is not None else None
        h = ctypes.c_void_p()
        c = c_config(cfg)
        _lib.check(_lib.lib().pmx_mgpu_create(ctypes.byref(c), n_devices, arr, ctypes.byref(h)))
        return cls(h, cfg)

    @classmethod
    def one_rank(cls, cfg: PoseidonConfig, device: int, rank: int, world: int, uid: bytes) -> "DeviceGroup":
        assert len(uid) == _lib.UNIQUE_ID_BYTES
        buf = (ctypes.c_uint8 * _lib.UNIQUE_ID_BYTES).from_buffer_copy(uid)
        h = ctypes.c_void_p()
        c = c_config(cfg)
        _lib.check(_lib.lib().pmx_mgpu_create_rank(ctypes.byref(c), device, rank, world, buf, ctypes.byref(h)))
        return cls(h, cfg)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().pmx_mgpu_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        i = _lib.PmxMgpuInfo()
        _lib.check(_lib.lib().pmx_mgpu_get_info(self._h, ctypes.byref(i)))
        v = i.rccl_version
        return {"world": i.world, "n_local": i.n_local, "first_rank": i.first_rank, "width": i.width,
                "rccl_version": v, "rccl_version_str": f"{v // 10000}.{(v // 100) % 100}.{v % 100}",
                "comm_ranks": i.comm_ranks, "comm_first_rank": i.comm_first_rank,
                "devices": [i.devices[k] for k in range(i.n_local)]}

    def stream(self, local: int = 0) -> int:
        """hipStream_t (as an integer) the group enqueues local device `local`'s work on."""
        return int(_lib.lib().pmx_mgpu_stream(self._h, local) or 0)

    def context(self, local: int = 0) -> Context:
        """The group's context of local device `local` (owned by the group): for the single-device *_dev entry points -
        hash, absorb, squeeze - on this shard, enqueued on self.stream(local)."""
        h = _lib.lib().pmx_mgpu_ctx(self._h, local)
        if not h:
            raise IndexError(local)
        return Context.borrowed(self.cfg, self.devices[local], h)

    def local_span(self, n_total: int, local: int) -> Tuple[int, int]:
        return shard_bounds(n_total, self.world, self.first_rank + local)

    def synchronize(self) -> None:
        _lib.check(_lib.lib().pmx_mgpu_synchronize(self._h))

    # ---- host batch ------------------------------------------------------------------------------------
    def permute_batch(self, states: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(states, dtype=np.uint64).copy()
        _lib.check(_lib.lib().pmx_mgpu_permute_batch(self._h, ctypes.c_void_p(out.ctypes.data), out.size // (self.cfg.t * 4)))
        return out

    def permute_batch_inplace(self, states: np.ndarray) -> None:
        assert states.dtype == np.uint64 and states.flags["C_CONTIGUOUS"]
        _lib.check(_lib.lib().pmx_mgpu_permute_batch(self._h, ctypes.c_void_p(states.ctypes.data),
                                                     states.size // (self.cfg.t * 4)))

    def hash_batch(self, msgs: np.ndarray, in_len: int, out_len: int, n: Optional[int] = None) -> np.ndarray:
        msgs = np.ascontiguousarray(msgs, dtype=np.uint64)
        if n is None:
            n = msgs.size // (in_len * 4)
        out = np.zeros((n, out_len, 4), dtype=np.uint64)
        _lib.check(_lib.lib().pmx_mgpu_hash_batch(self._h, ctypes.c_void_p(msgs.ctypes.data) if msgs.size else None, in_len,
                                                  ctypes.c_void_p(out.ctypes.data), out_len, n))
        return out

    def merkle_root(self, leaves: np.ndarray) -> np.ndarray:
        leaves = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, 4)
        root = np.zeros(4, dtype=np.uint64)
        _lib.check(_lib.lib().pmx_mgpu_merkle_2to1(self._h, ctypes.c_void_p(leaves.ctypes.data), leaves.shape[0],
                                                   ctypes.c_void_p(root.ctypes.data)))
        return root

    # ---- device-resident shards (raw device addresses, one per local device; only enqueue) -----------------
    def permute_shards_dev(self, d_shards: Sequence[int], n_total: int) -> None:
        _lib.check(_lib.lib().pmx_mgpu_permute_shards_dev(self._h, _ptr_array(d_shards), n_total))

    def all_gather_dev(self, d_shards: Sequence[int], d_all: Sequence[int], n_total: int, row_elems: int) -> None:
        _lib.check(_lib.lib().pmx_mgpu_all_gather_dev(self._h, _ptr_array(d_shards), _ptr_array(d_all), n_total, row_elems))

    def gather_dev(self, d_shards: Sequence[int], d_all: Sequence[int], n_total: int, row_elems: int, root: int) -> None:
        """the result to ONE rank (d_all entries of the other slots are not read: pass 0)"""
        _lib.check(_lib.lib().pmx_mgpu_gather_dev(self._h, _ptr_array(d_shards), _ptr_array(d_all), n_total, row_elems, root))

    def permute_gather_dev(self, d_shards: Sequence[int], d_all: Sequence[int], n_total: int, root: int = -1, chunks: int = 8) -> None:
        """the last step and its gather (root < 0: to every rank), the transfers of piece i behind piece i's kernel"""
        _lib.check(_lib.lib().pmx_mgpu_permute_gather_dev(self._h, _ptr_array(d_shards), _ptr_array(d_all), n_total, root, chunks))

    def merkle_2to1_dev(self, d_nodes: Sequence[int], d_top: Sequence[int], n_leaves: int) -> None:
        _lib.check(_lib.lib().pmx_mgpu_merkle_2to1_dev(self._h, _ptr_array(d_nodes), _ptr_array(d_top), n_leaves))
