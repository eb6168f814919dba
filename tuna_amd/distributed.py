"""N > 1 GPUs: one process per GPU, each holding a shard of the (ij) shell-pair rows of the ERI tensor; a Fock build is
the local J/K pass followed by ONE all-reduce of the stacked [J;K].  torch.distributed is the
transport: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests."""
from __future__ import annotations

import os

import numpy as np

from . import _lib


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def packed_pad() -> int:
    """Alignment unit of the packed layout in doubles (tf_packed_pad)."""
    return int(_lib.lib().tf_packed_pad())


def packed_tri_offset(k, pad=None):
    """First padded index of row k of the (k >= l) triangle: the row lengths 1, 2, 3, ... each rounded up to the alignment
    unit (tf_jkpacked.hip.h: tri_off)."""
    pad = packed_pad() if pad is None else int(pad)
    k = np.asarray(k, dtype=np.int64)
    q, r = k // pad, k % pad
    return pad * (pad * q * (q + 1) // 2 + r * (q + 1))


def packed_row_length(i, j, pad=None):
    """Stored doubles of tensor row (i >= j) in the packed layout: the pairs (k,l) <= (i,j), rounded up to the unit."""
    pad = packed_pad() if pad is None else int(pad)
    return (packed_tri_offset(i, pad) + np.asarray(j, dtype=np.int64) + pad) & ~np.int64(pad - 1)


def shell_pair_rows(shells, spherical: bool = True, layout: str = "packed") -> np.ndarray:
    """Weight of every bra shell pair (A >= B, A-major) in the shard plan of tf_build_eri: the stored elements of its rows
    (packed layout), or its row count (rows layout, every row has the same length)."""
    dim = [(s.n_sph if spherical else s.n_cart) for s in shells]
    off = np.concatenate([[0], np.cumsum(dim)]).astype(np.int64)
    w = []
    for A in range(len(shells)):
        for B in range(A + 1):
            i, j = np.meshgrid(np.arange(off[A], off[A + 1]), np.arange(off[B], off[B + 1]), indexing="ij")
            keep = i >= j
            w.append(int(packed_row_length(i, j)[keep].sum()) if layout == "packed" else int(keep.sum()))
    return np.asarray(w, dtype=np.int64)


def shard_owner(shells, world: int, spherical: bool = True, layout: str = "packed") -> np.ndarray:
    """owner[p] = rank that generates and keeps the rows of shell pair p (A >= B, A-major) -- the library's own plan
    (tf_shard_plan_pairs): for every A the B range is cut into `world` contiguous segments of equal weight."""
    dim = np.asarray([(s.n_sph if spherical else s.n_cart) for s in shells], dtype=np.int32)
    owner = np.zeros(len(dim) * (len(dim) + 1) // 2, dtype=np.int32)
    rc = _lib.lib().tf_shard_plan_pairs(len(dim), _lib.ptr(dim), 1 if layout == "packed" else 0, int(world), _lib.ptr(owner))
    if rc != 0:
        raise _lib.TunaError("tf_shard_plan_pairs failed", rc)
    return owner


def row_owner_matrix(shells, world: int, spherical: bool = True, layout: str = "packed") -> np.ndarray:
    """owner[i, j] (i >= j) of every AO-pair row; -1 above the diagonal."""
    owner = shard_owner(shells, world, spherical, layout)
    dim = [(s.n_sph if spherical else s.n_cart) for s in shells]
    off = np.concatenate([[0], np.cumsum(dim)])
    N = int(off[-1])
    out = np.full((N, N), -1, dtype=np.int32)
    p = 0
    for A in range(len(shells)):
        for B in range(A + 1):
            out[off[A]:off[A + 1], off[B]:off[B + 1]] = owner[p]
            p += 1
    out[np.triu_indices(N, 1)] = -1
    return out


def all_reduce_jk_(jk, group=None):
    """In-place sum over ranks of a stacked [2, N, N] (or [n_dens, 2, N, N]) tensor: the single exchange step of a build."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(jk, group=group)
    return jk


class ShardedFock:
    """Fock builds over a sharded tensor: engine.fock_jk_device on this rank's rows + all-reduce."""

    def __init__(self, engine, device=None):
        import torch
        self.engine = engine
        self.device = device if device is not None else torch.device("cuda", engine.device)
        N = engine.N
        self._P = torch.zeros((N, N), dtype=torch.float64, device=self.device)
        self._JK = torch.zeros((2, N, N), dtype=torch.float64, device=self.device)

    def __call__(self, P: np.ndarray):
        import torch
        self._P.copy_(torch.from_numpy(np.ascontiguousarray(P, dtype=np.float64)))
        stream = torch.cuda.current_stream().cuda_stream
        self.engine.fock_jk_device(self._P.data_ptr(), self._JK[0].data_ptr(), self._JK[1].data_ptr(), 1, stream)
        all_reduce_jk_(self._JK)
        out = self._JK.cpu().numpy()
        return out[0], out[1]
