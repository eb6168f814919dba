"""Drop-in face of the reference's extension module `tuna_integrals.tuna_integral` (seam 1).

Same names, argument meaning and error behaviour as pyx:78-140 (`Basis`), pyx:282 (`calculate_one_electron_integrals`),
pyx:626 (`calculate_cross_basis_overlap_matrix`), pyx:1267 (`calculate_electron_repulsion_integrals`) and pyx:1376
(`calculate_electron_repulsion_integral`) -- every number comes from the HIP library through the C ABI
(include/tunafock.h); `num_threads` is accepted and ignored (the grid is the GPU's).  A TUNA checkout switches with
`from tuna_amd import integral as ints`.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import TunaError, f64, ptr
from .engine import Engine
from .molecule import AOList

_default_engine: Engine | None = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine(0)
    return _default_engine


class Basis:
    """One Cartesian AO: contracted Gaussian with angular exponents `shell` = (l, m, n) at `origin` (pyx:78-234).

    `coefs` are normalised in place on construction and `norm` holds the primitive norms, exactly as
    `Basis.normalize` (pyx:174-210) leaves them; every integral uses `norm[i] * coefs[i]`."""

    __slots__ = ("origin", "shell", "num_exps", "exps", "coefs", "norm", "raw_coefs")

    def __init__(self, origin, shell, num_exps, exps, coefs):
        self.origin = np.array(origin, dtype=np.float64).reshape(3)
        self.shell = np.array(shell, dtype=np.int64).reshape(3)
        self.num_exps = int(num_exps)
        self.exps = np.array(exps, dtype=np.float64).reshape(self.num_exps)
        self.raw_coefs = np.array(coefs, dtype=np.float64).reshape(self.num_exps)
        self.coefs = self.raw_coefs.copy()
        self.norm = np.zeros(self.num_exps)
        rc = _lib.lib().tf_normalize(int(self.shell[0]), int(self.shell[1]), int(self.shell[2]), self.num_exps,
                                     ptr(self.exps), ptr(self.coefs), ptr(self.norm))
        if rc != 0:
            raise TunaError("Basis set malformed! If using a custom basis set, check the file format carefully.", rc)


def aos_from_basis_list(bfs) -> AOList:
    n = len(bfs)
    nprim = np.array([b.num_exps for b in bfs], dtype=np.int32)
    off = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(nprim, out=off[1:])
    return AOList(np.array([b.origin for b in bfs], dtype=np.float64).reshape(n, 3),
                  np.array([b.shell for b in bfs], dtype=np.int32).reshape(n, 3), nprim, off,
                  np.concatenate([b.exps for b in bfs]), np.concatenate([b.raw_coefs for b in bfs]))


def calculate_one_electron_integrals(n_basis, basis_functions, n_atoms, atoms, dipole_origin, num_threads=0):
    """-> (S_cart, T_cart, V_cart, D_cart[3], Q_cart[3]); atoms need `.origin` and `.charge` (pyx:282-435)."""
    eng = default_engine()
    eng.set_basis(aos_from_basis_list(basis_functions[:n_basis]))
    xyz = [np.asarray(a.origin, dtype=np.float64) for a in atoms[:n_atoms]]
    chg = [float(a.charge) for a in atoms[:n_atoms]]
    return eng.one_electron(xyz, chg, f64(dipole_origin), spherical=False)


def calculate_cross_basis_overlap_matrix(n_basis_1, n_basis_2, basis_functions_1, basis_functions_2, num_threads=0):
    eng = default_engine()
    eng.set_basis(aos_from_basis_list(basis_functions_1[:n_basis_1]))
    return eng.cross_overlap(aos_from_basis_list(basis_functions_2[:n_basis_2]))


def calculate_electron_repulsion_integrals(n_basis, ERI_AO, bfs, num_threads=0):
    """Fills the caller's float64[n,n,n,n] with (ij|kl), all 8 images, exact zeros where x/y parity is odd (pyx:1267-1355)."""
    ERI_AO = np.asarray(ERI_AO)
    if ERI_AO.shape != (n_basis,) * 4 or ERI_AO.dtype != np.float64 or not ERI_AO.flags.c_contiguous:
        raise TunaError("ERI_AO must be a C-contiguous float64 array of shape (n_basis,)*4")
    eng = default_engine()
    eng.set_basis(aos_from_basis_list(bfs[:n_basis])).build_eri(spherical=False)
    eng.copy_eri(ERI_AO)
    return ERI_AO


def calculate_electron_repulsion_integral(bf_1, bf_2, bf_3, bf_4) -> float:
    return default_engine().eri_element(aos_from_basis_list([bf_1, bf_2, bf_3, bf_4]))
