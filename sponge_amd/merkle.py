"""2-to-1 Poseidon Merkle trees on the GPU: the natural consumer of the compression mode (SURVEY section 8(f) rank 4;
the container itself lives upstream in ark-crypto-primitives, not in arkworks-rs/sponge).

A parent is  (PoseidonSponge::new; absorb([left, right]); squeeze_native_field_elements(1))[0]
(reference src/poseidon/mod.rs:219-254, 321-341).  Nodes are kept as one array [2m-1][4]: leaves, then every level,
root last - the layout pmx_merkle_2to1 produces.  Authentication paths are verified in batch: all paths advance
one level per hash_batch call."""
from __future__ import annotations

from typing import List

import numpy as np

import ctypes

from . import _lib
from .poseidon import PoseidonConfig


class MerkleTree:
    def __init__(self, parameters: PoseidonConfig, leaves: np.ndarray, device: int = 0):
        leaves = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, 4)
        m = leaves.shape[0]
        assert m >= 1 and (m & (m - 1)) == 0, "number of leaves must be a power of two"
        self.parameters = parameters
        self.device = device
        self.n_leaves = m
        self.depth = m.bit_length() - 1
        if m == 1:
            self.nodes, self._root = leaves.copy(), leaves[0].copy()
        else:
            self.nodes, self._root = parameters.context(device).merkle_2to1(leaves)

    @property
    def root(self) -> np.ndarray:
        return self._root

    def level_offset(self, level: int) -> int:
        """Index of the first node of `level` (0 = leaves) in `nodes`."""
        return 2 * self.n_leaves - (self.n_leaves >> (level - 1) if level else 2 * self.n_leaves)

    def path(self, leaf_index: int) -> np.ndarray:
        """Sibling of the leaf, then of each ancestor, bottom-up: [depth][4]."""
        assert 0 <= leaf_index < self.n_leaves
        return self.paths([leaf_index])[0]

    def paths(self, leaf_indices) -> np.ndarray:
        """[k][depth][4] for k leaves (pmx_merkle_paths: a host-side gather over the node array)."""
        idx = np.ascontiguousarray(leaf_indices, dtype=np.uint64)
        out = np.zeros((idx.shape[0], self.depth, 4), dtype=np.uint64)
        nodes = np.ascontiguousarray(self.nodes, dtype=np.uint64)
        _lib.check(_lib.lib().pmx_merkle_paths(ctypes.c_void_p(nodes.ctypes.data), self.n_leaves, ctypes.c_void_p(idx.ctypes.data),
                                               idx.shape[0], ctypes.c_void_p(out.ctypes.data)))
        return out


def verify_paths(parameters: PoseidonConfig, leaves: np.ndarray, indices, paths: np.ndarray, root: np.ndarray,
                 device: int = 0) -> np.ndarray:
    """k authentication paths at once: leaves [k][4], indices [k], paths [k][depth][4] -> bool[k]
    (pmx_merkle_verify_paths: one upload, one device step per level, one download; an index with bits at or above
    `depth` names no leaf and verifies as False)."""
    cur = np.ascontiguousarray(leaves, dtype=np.uint64).reshape(-1, 4)
    idx = np.ascontiguousarray(indices, dtype=np.uint64)
    paths = np.ascontiguousarray(paths, dtype=np.uint64)
    k = cur.shape[0]
    depth = paths.shape[1] if paths.ndim == 3 else 0
    root = np.ascontiguousarray(root, dtype=np.uint64).reshape(4)
    ok = np.zeros(k, dtype=np.uint8)
    ctx = parameters.context(device)
    _lib.check(_lib.lib().pmx_merkle_verify_paths(ctx._h, ctypes.c_void_p(cur.ctypes.data), ctypes.c_void_p(idx.ctypes.data),
                                                  ctypes.c_void_p(paths.ctypes.data) if paths.size else None, depth, k,
                                                  ctypes.c_void_p(root.ctypes.data), ctypes.c_void_p(ok.ctypes.data)))
    return ok.astype(bool)
