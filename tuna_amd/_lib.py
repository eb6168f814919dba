"""ctypes binding of libtunafock.so (include/tunafock.h).  No CPU fallback: if the library or a GPU is
missing, every compute call raises TunaError."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TUNAFOCK_LIB") or os.path.join(HERE, "libtunafock.so")


class TunaError(RuntimeError):
    """Mirror of the reference's TunaError raised by error() (tuna_util.py:933-944)."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


class ScfOpts(C.Structure):
    _fields_ = [("max_iter", C.c_int32), ("use_diis", C.c_int32), ("max_diis", C.c_int32), ("damping", C.c_int32),
                ("damping_factor", C.c_double), ("max_damping", C.c_double), ("conv_delta_E", C.c_double),
                ("conv_max_DP", C.c_double), ("conv_rms_DP", C.c_double), ("conv_commutator", C.c_double),
                ("hfx", C.c_double), ("n_atom_ao", C.c_int32 * 2), ("n_atoms", C.c_int32)]


class ScfResult(C.Structure):
    _fields_ = [("energy", C.c_double), ("components", C.c_double * 7), ("n_iter", C.c_int32), ("converged", C.c_int32),
                ("P", C.c_void_p), ("C", C.c_void_p), ("eps", C.c_void_p), ("F", C.c_void_p), ("table", C.c_void_p),
                ("fock_seconds", C.c_double), ("eig_seconds", C.c_double), ("wall_seconds", C.c_double)]


class ScfUhfResult(C.Structure):                          # tf_scf_uhf_result
    _fields_ = [("common", ScfResult), ("P_spin", C.c_void_p * 2), ("C_spin", C.c_void_p * 2), ("eps_spin", C.c_void_p * 2),
                ("F_spin", C.c_void_p * 2)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)     # tf_allreduce_fn

EXPORTS = ["tf_create", "tf_destroy", "tf_last_error", "tf_version", "tf_normalize", "tf_set_basis", "tf_get_norms",
           "tf_dims", "tf_get_sph_matrix", "tf_one_electron", "tf_cross_overlap", "tf_build_eri", "tf_eri_storage",
           "tf_copy_eri", "tf_sample_eri", "tf_eri_element", "tf_fock_jk", "tf_fock_jk_device", "tf_scf_rhf", "tf_scf_uhf",
           "tf_orthogonaliser", "tf_eri_timings", "tf_eri_counts", "tf_shard_plan", "tf_jk_profile",
           "tf_jk_profile_read", "tf_diagonalise", "tf_eigh_probe", "tf_eigh_stats", "tf_jk_path_stats", "tf_ao_to_mo", "tf_mp2_rhf", "tf_dft_setup", "tf_dft_vxc",
           "tf_dft_clear", "tf_set_eri_layout", "tf_eri_layout", "tf_shard_plan_pairs", "tf_packed_pad", "tf_eri_flops", "tf_segment_pad", "tf_set_allreduce", "tf_scf_rhf_batch",
           "tf_comm_unique_id", "tf_comm_init", "tf_comm_destroy", "tf_comm_attached"]

_lib = None


def build_library(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(HERE, "csrc")
    # force: `make -B` recompiles every object (a library newer than its sources is not trusted: the driver's build check)
    subprocess.check_call(["make", "-j4", "-C", src_dir, "--no-print-directory"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TunaError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(tuna_amd has no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, ci, cd = C.c_void_p, C.c_int, C.c_double
    dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.tf_create.restype = vp; L.tf_create.argtypes = [ci, ci, ci]
    L.tf_destroy.restype = None; L.tf_destroy.argtypes = [vp]
    L.tf_last_error.restype = C.c_char_p; L.tf_last_error.argtypes = [vp]
    L.tf_version.restype = ci; L.tf_version.argtypes = []
    L.tf_normalize.restype = ci; L.tf_normalize.argtypes = [ci, ci, ci, ci, vp, vp, vp]
    L.tf_set_basis.restype = ci; L.tf_set_basis.argtypes = [vp, ci, vp, vp, vp, vp, vp]
    L.tf_get_norms.restype = ci; L.tf_get_norms.argtypes = [vp, vp, vp]
    L.tf_dims.restype = ci; L.tf_dims.argtypes = [vp, ip, ip, ip]
    L.tf_get_sph_matrix.restype = ci; L.tf_get_sph_matrix.argtypes = [vp, vp]
    L.tf_one_electron.restype = ci; L.tf_one_electron.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, vp]
    L.tf_cross_overlap.restype = ci; L.tf_cross_overlap.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp]
    L.tf_build_eri.restype = ci; L.tf_build_eri.argtypes = [vp, ci]
    L.tf_set_eri_layout.restype = ci; L.tf_set_eri_layout.argtypes = [vp, ci]
    L.tf_eri_layout.restype = ci; L.tf_eri_layout.argtypes = [vp]
    L.tf_packed_pad.restype = ci; L.tf_packed_pad.argtypes = []
    L.tf_eri_storage.restype = ci; L.tf_eri_storage.argtypes = [vp, lp, lp, ip, ip]
    L.tf_eri_flops.restype = ci; L.tf_eri_flops.argtypes = [vp, vp]
    L.tf_segment_pad.restype = ci; L.tf_segment_pad.argtypes = []
    L.tf_set_allreduce.restype = ci; L.tf_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    L.tf_comm_unique_id.restype = ci; L.tf_comm_unique_id.argtypes = [vp]
    L.tf_comm_init.restype = ci; L.tf_comm_init.argtypes = [vp, vp, ci, ci]
    L.tf_comm_destroy.restype = ci; L.tf_comm_destroy.argtypes = [vp]
    L.tf_comm_attached.restype = ci; L.tf_comm_attached.argtypes = [vp]
    L.tf_copy_eri.restype = ci; L.tf_copy_eri.argtypes = [vp, vp]
    L.tf_sample_eri.restype = ci; L.tf_sample_eri.argtypes = [vp, C.c_int64, vp, vp]
    L.tf_eri_element.restype = ci; L.tf_eri_element.argtypes = [vp, vp, vp, vp, vp, vp, dp]
    L.tf_fock_jk.restype = ci; L.tf_fock_jk.argtypes = [vp, ci, vp, vp, vp]
    L.tf_fock_jk_device.restype = ci; L.tf_fock_jk_device.argtypes = [vp, ci, vp, vp, vp, vp]
    L.tf_scf_rhf.restype = ci
    L.tf_scf_rhf.argtypes = [vp, C.POINTER(ScfOpts), vp, vp, vp, vp, vp, vp, cd, ci, cd, C.POINTER(ScfResult)]
    L.tf_scf_rhf_batch.restype = ci; L.tf_scf_rhf_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, ci, cd, vp, vp, vp]
    L.tf_scf_uhf.restype = ci
    L.tf_scf_uhf.argtypes = [vp, C.POINTER(ScfOpts), vp, vp, vp, vp, vp, vp, vp, cd, ci, ci, cd, C.POINTER(ScfUhfResult)]
    L.tf_orthogonaliser.restype = ci; L.tf_orthogonaliser.argtypes = [vp, ci, vp, vp, vp, dp]
    L.tf_eri_timings.restype = ci; L.tf_eri_timings.argtypes = [vp, vp]
    L.tf_eri_counts.restype = ci; L.tf_eri_counts.argtypes = [vp, vp]
    L.tf_shard_plan.restype = ci; L.tf_shard_plan.argtypes = [ci, vp, ci, vp]
    L.tf_shard_plan_pairs.restype = ci; L.tf_shard_plan_pairs.argtypes = [ci, vp, ci, ci, vp]
    L.tf_diagonalise.restype = ci; L.tf_diagonalise.argtypes = [vp, ci, vp, vp, vp, vp]
    L.tf_ao_to_mo.restype = ci; L.tf_ao_to_mo.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp]
    L.tf_mp2_rhf.restype = ci; L.tf_mp2_rhf.argtypes = [vp, ci, ci, vp, vp, dp, dp, dp]
    L.tf_dft_setup.restype = ci; L.tf_dft_setup.argtypes = [vp, C.c_int64, vp, vp, ci, ci, cd, cd, cd]
    L.tf_dft_vxc.restype = ci; L.tf_dft_vxc.argtypes = [vp, vp, vp, dp, dp, dp]
    L.tf_dft_clear.restype = ci; L.tf_dft_clear.argtypes = [vp]
    L.tf_eigh_probe.restype = ci; L.tf_eigh_probe.argtypes = [vp, ci, ci, ci, dp]
    L.tf_eigh_stats.restype = ci; L.tf_eigh_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.tf_jk_path_stats.restype = ci; L.tf_jk_path_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.tf_jk_profile.restype = ci; L.tf_jk_profile.argtypes = [vp, ci]
    L.tf_jk_profile_read.restype = ci; L.tf_jk_profile_read.argtypes = [vp, dp, lp]
    _lib = L
    return L


def ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)
