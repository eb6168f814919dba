"""TEST INFRASTRUCTURE ONLY -- Python face of the CPU oracle.

Loads oracle/tuna_oracle.c (compiled on demand with gcc into oracle/_build/) and, when present,
the compiled *unmodified* reference engine from oracle/_ref/ (see oracle/build_ref.sh).  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (tuna_amd/) never does.

Parity status: PINNED against the reference engine and its golden vectors (tests/test_oracle.py:
test_oracle_matches_compiled_reference_on_random_basis runs the compiled engine of oracle/_ref beside the
restatement where /root/reference exists; the other tests replay tests/golden/, which that engine produced).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(HERE, "tuna_oracle.c")
_LIB = os.path.join(HERE, "_build", "libtunaoracle.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(_SRC):
        os.makedirs(os.path.dirname(_LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fPIC", "-shared", "-o", _LIB, _SRC, "-lm"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_normalize.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
        L.orc_normalize.restype = None
        L.orc_boys.argtypes = [C.c_int, C.c_double]
        L.orc_boys.restype = C.c_double
        L.orc_one_electron.argtypes = [C.c_int, _dp, _ip, _ip, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp,
                                       _dp, _dp, _dp, _dp, _dp, C.c_int]
        L.orc_one_electron.restype = None
        L.orc_cross_overlap.argtypes = [C.c_int, _dp, _ip, _ip, _dp, _dp, _dp,
                                        C.c_int, _dp, _ip, _ip, _dp, _dp, _dp, _dp]
        L.orc_cross_overlap.restype = None
        L.orc_eri.argtypes = [C.c_int, _dp, _ip, _ip, _dp, _dp, _dp, _dp, C.c_int]
        L.orc_eri.restype = None
        L.orc_eri_element.argtypes = [_dp, _ip, _ip, _dp, _dp, _dp]
        L.orc_eri_element.restype = C.c_double
        _lib = L
    return _lib


def normalize(aos):
    """Per-AO primitive norms and normalised coefficients (pyx:174-210).  Returns (norm, coefs_n)."""
    L = lib()
    norm = np.zeros_like(aos.exps)
    coefs = aos.coefs.copy()
    for i in range(aos.n):
        a, b = int(aos.prim_off[i]), int(aos.prim_off[i + 1])
        e = np.ascontiguousarray(aos.exps[a:b])
        c = np.ascontiguousarray(coefs[a:b])
        nrm = np.zeros(b - a)
        L.orc_normalize(int(aos.lmn[i, 0]), int(aos.lmn[i, 1]), int(aos.lmn[i, 2]), b - a, e, c, nrm)
        coefs[a:b] = c
        norm[a:b] = nrm
    return norm, coefs


def one_electron(aos, atom_xyz, atom_charge, dipole_origin, threads: int = 0):
    """S, T, V, D[3], Q[3] over Cartesian AOs (pyx:282-435)."""
    norm, coefs = normalize(aos)
    n = aos.n
    S = np.zeros((n, n)); T = np.zeros((n, n)); V = np.zeros((n, n))
    D = np.zeros((3, n, n)); Q = np.zeros((3, n, n))
    xyz = np.ascontiguousarray(atom_xyz, dtype=np.float64).reshape(-1, 3)
    chg = np.ascontiguousarray(atom_charge, dtype=np.float64)
    lib().orc_one_electron(n, aos.origin, aos.lmn, aos.prim_off, aos.exps, coefs, norm, len(chg), xyz, chg,
                           np.ascontiguousarray(dipole_origin, dtype=np.float64), S, T, V, D, Q,
                           threads or os.cpu_count())
    return S, T, V, D, Q


def cross_overlap(aos1, aos2):
    n1, c1 = normalize(aos1)
    n2, c2 = normalize(aos2)
    S = np.zeros((aos1.n, aos2.n))
    lib().orc_cross_overlap(aos1.n, aos1.origin, aos1.lmn, aos1.prim_off, aos1.exps, c1, n1,
                            aos2.n, aos2.origin, aos2.lmn, aos2.prim_off, aos2.exps, c2, n2, S)
    return S


def eri(aos, threads: int = 0):
    """Dense Cartesian (ij|kl) tensor with all 8 images (pyx:1267-1355)."""
    norm, coefs = normalize(aos)
    n = aos.n
    out = np.empty((n, n, n, n))
    lib().orc_eri(n, aos.origin, aos.lmn, aos.prim_off, aos.exps, coefs, norm, out, threads or os.cpu_count())
    return out


def boys(m: int, T: float) -> float:
    return lib().orc_boys(int(m), float(T))


# ---- the real reference engine (oracle/_ref), when it has been built --------------------------

def ref_engine():
    """Import the compiled reference `tuna_integrals.tuna_integral`, or return None if it is not built."""
    root = os.path.join(HERE, "_ref")
    if not os.path.isdir(os.path.join(root, "tuna_integrals")):
        return None
    if root not in sys.path:
        sys.path.insert(0, root)
    try:
        from tuna_integrals import tuna_integral  # type: ignore
        return tuna_integral
    except Exception:
        return None


def ref_basis_list(aos):
    ints = ref_engine()
    out = []
    for i in range(aos.n):
        a, b = int(aos.prim_off[i]), int(aos.prim_off[i + 1])
        out.append(ints.Basis(aos.origin[i].copy(), aos.lmn[i].astype(np.int64), b - a,
                              [float(x) for x in aos.exps[a:b]], [float(x) for x in aos.coefs[a:b]]))
    return out


class _RefAtom:
    def __init__(self, origin, charge):
        self.origin = np.asarray(origin, dtype=np.float64)
        self.charge = charge


def ref_one_electron(aos, atom_xyz, atom_charge, dipole_origin, threads=8):
    ints = ref_engine()
    bfs = ref_basis_list(aos)
    atoms = [_RefAtom(x, c) for x, c in zip(np.asarray(atom_xyz).reshape(-1, 3), atom_charge)]
    S, T, V, D, Q = ints.calculate_one_electron_integrals(aos.n, bfs, len(atoms), atoms,
                                                          np.asarray(dipole_origin, dtype=np.float64), threads)
    return tuple(np.asarray(x) for x in (S, T, V, D, Q))


def ref_eri(aos, threads=8):
    ints = ref_engine()
    bfs = ref_basis_list(aos)
    out = np.empty((aos.n,) * 4)
    ints.calculate_electron_repulsion_integrals(aos.n, out, bfs, threads)
    return out
