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


class _DeviceBuffer:
    """A device allocation owned by libtunafock, exposed through the CUDA array interface so that torch can alias it."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def allreduce_with_status(t, group=None, backend=None, to_host=None, from_host=None):
    """The exchange step on a buffer whose LAST element is a status word (0 on entry): sum-all-reduce of the whole buffer.  `to_host` /
    `from_host` stage the buffer through the host (gloo).  A staging failure on this rank does not skip the collective -- the other
    ranks would wait in it for ever -- but sends zeros with the status word set, so that every rank sees a non-zero status afterwards.
    Returns 0, or 1 when this rank could not even take part (nothing to be done about that here)."""
    import torch
    import torch.distributed as dist
    backend = backend or dist.get_backend(group)
    if backend == "nccl":
        dist.all_reduce(t, group=group)                      # on the current stream: the caller selects the library's stream
        return 0
    try:
        h = (to_host or (lambda x: x.cpu()))(t)
    except Exception:
        import traceback
        traceback.print_exc()
        h = torch.zeros(t.numel(), dtype=torch.float64)
        h[-1] = 1.0
    dist.all_reduce(h, group=group)
    try:
        (from_host or (lambda x, y: x.copy_(y)))(t, h)
    except Exception:
        import traceback
        traceback.print_exc()
        return 1
    return 0


def attach_allreduce(engine, group=None):
    """Registers the exchange step of a sharded Fock build with the library (tf_set_allreduce): the native SCF cycles then run
    on a tensor spread over the ranks of `group` -- per iteration ONE sum-all-reduce of the stacked partial [J;K] (plus a status word,
    and a 16-double agreement vector per iteration), on the device buffer the library hands over.  Backend "nccl" (= RCCL over xGMI):
    the collective is issued on the stream the library passes (`torch.cuda.ExternalStream`; the legacy default stream when it passes
    NULL), i.e. ordered after the library's copies into the buffer and before its copies out of it, whatever torch's current stream
    is.  Backend "gloo" (CPU tests, several ranks sharing one card): the buffer is staged through the host."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise _lib.TunaError("attach_allreduce: torch.distributed is not initialised (one process per GPU, backend nccl)")
    backend = dist.get_backend(group)
    device = torch.device("cuda", engine.device)

    def hook(user, buf, count, stream):
        try:
            t = torch.as_tensor(_DeviceBuffer(buf, count), device=device)
            if backend == "nccl":
                s = torch.cuda.ExternalStream(int(stream), device=device) if stream else torch.cuda.default_stream(device)
                with torch.cuda.stream(s):
                    return allreduce_with_status(t, group, backend)

            def to_host(x):
                torch.cuda.synchronize(device)
                return x.cpu()

            def from_host(x, h):
                x.copy_(h)
                torch.cuda.synchronize(device)
            return allreduce_with_status(t, group, backend, to_host, from_host)
        except Exception:                                   # no exception may cross the C ABI
            import traceback
            traceback.print_exc()
            return 1

    engine.set_allreduce(hook)
    return engine


def attach_rccl(engine, group=None):
    """The exchange step INSIDE the library (tf_comm_init): one RCCL communicator over the ranks of `group` (torch.distributed is used
    once, to hand rank 0's unique id to the others); afterwards every Fock build of the native cycles, tf_fock_jk_device and the AO->MO
    transformation issue ncclAllReduce themselves on the library's stream -- no Python frame per build."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise _lib.TunaError("attach_rccl: torch.distributed is not initialised (one process per GPU)")
    rank, size = dist.get_rank(group), dist.get_world_size(group)
    box = [type(engine).comm_unique_id() if rank == 0 else None]
    if size > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    engine.comm_init(box[0], rank, size)
    return engine


class ShardedFock:
    """Fock builds over a sharded tensor: engine.fock_jk_device on this rank's rows + ONE all-reduce of the stacked [J;K];
    one [N,N] density or several [n,N,N] (a UHF build passes alpha and beta)."""

    def __init__(self, engine, device=None):
        import torch
        self.engine = engine
        self.device = device if device is not None else torch.device("cuda", engine.device)
        self._bufs = {}

    def _buffers(self, nd):
        import torch
        if nd not in self._bufs:
            N = self.engine.N
            self._bufs[nd] = (torch.zeros((nd, N, N), dtype=torch.float64, device=self.device),
                              torch.zeros((2, nd, N, N), dtype=torch.float64, device=self.device))
        return self._bufs[nd]

    def __call__(self, P: np.ndarray):
        import torch
        import torch.distributed as dist
        P = np.ascontiguousarray(P, dtype=np.float64)
        single = P.ndim == 2
        Pn = P[None] if single else P
        dP, dJK = self._buffers(Pn.shape[0])
        dP.copy_(torch.from_numpy(Pn))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.engine.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), Pn.shape[0], stream)
        if self.engine.comm_attached():                    # summed over the ranks inside the library (tf_comm_init)
            out = dJK.cpu().numpy()
        elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl":
                dist.all_reduce(dJK)
                out = dJK.cpu().numpy()
            else:                                           # gloo: reduce on the host
                h = dJK.cpu()
                dist.all_reduce(h)
                out = h.numpy()
        else:
            out = dJK.cpu().numpy()
        return (out[0, 0], out[1, 0]) if single else (out[0], out[1])
