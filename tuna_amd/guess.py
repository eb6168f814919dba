"""Initial guess for the SCF: superposition of atomic densities (the reference's default)
and the core-Hamiltonian guess.

Reference: form_minimal_basis_superposition_density tuna_guess.py:90-108, project_density_matrix :209-236,
calculate_superposition_guess :247-299, setup_initial_guess :363-433 (guess energy = H_core . P, :429),
enforce_density_matrix_idempotency tuna_kernel.py:112-141 -> clean_density_matrix tuna_dft.py:35-41.
The tabulated spherically averaged HF/STO-3G atomic densities (tuna_util.py:1676-1924) are data, shipped as
tuna_amd/data/atomic_data.json.  The one integral step -- the STO-3G x target cross overlap -- runs on the GPU
(tf_cross_overlap); the rest is O(N^2 n_min) host algebra exactly as in the reference.
"""
from __future__ import annotations

import json
import os
from functools import lru_cache

import numpy as np

from . import molecule as mol
from ._lib import TunaError

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "atomic_data.json")


@lru_cache(maxsize=1)
def atomic_data() -> dict:
    with open(_DATA) as f:
        return json.load(f)


def atomic_mass(symbol: str) -> float:
    s = symbol.upper()
    s = s[1:] if s.startswith("X") and s[1:] in atomic_data() else s
    return float(atomic_data()[s]["mass"])


def centre_of_mass(atoms) -> float:
    """tuna_util.py:775-793: sum_i m_i * (x_i + y_i + z_i) / sum_i m_i -- on the z axis this is the z coordinate."""
    m = np.array([atomic_mass(a.symbol) for a in atoms])
    xyz = np.array([a.origin for a in atoms])
    return float(np.einsum("i,ij->", m, xyz) / np.sum(m))


def form_minimal_basis_superposition_density(atoms) -> np.ndarray:        # tuna_guess.py:90-108
    from scipy.linalg import block_diag
    dens = []
    for a in atoms:
        d = atomic_data()[a.symbol.upper().lstrip("X") if a.symbol.upper().startswith("X") else a.symbol.upper()]["density"]
        if a.charge == 0 or d is None:
            raise TunaError("superposition-of-atomic-densities guess needs real atoms (ghost atom present: use COREGUESS)")
        dens.append(np.array(d, dtype=float))
    return block_diag(dens[0], dens[1]) / 2 if len(dens) > 1 else dens[0]


def project_density_matrix(P_to_project, S_cross, S_target_inverse, U):     # tuna_guess.py:209-236
    S_cross = U @ S_cross
    X = S_target_inverse @ S_cross
    return X @ P_to_project @ X.T


def clean_density_matrix(P, S, n_electrons):                                # tuna_dft.py:35-41
    scale = n_electrons / np.trace(P @ S) if n_electrons > 0 else 0
    return P * scale


def superposition_guess(engine, atoms, S, S_inverse, U, n_alpha, n_beta, H_core):
    """(P, P_alpha, P_beta, E_guess) of the reference's default SAD guess for the basis currently set on `engine`."""
    P_minimal = form_minimal_basis_superposition_density(atoms)
    minimal = mol.expand_cartesian_aos(mol.build_shells(atoms, "STO-3G"))
    S_cross = engine.cross_overlap(minimal)                                 # Cartesian target x STO-3G, on the GPU
    P_spin = project_density_matrix(P_minimal, S_cross, S_inverse, U)
    P_alpha = clean_density_matrix(P_spin, S, n_alpha)                      # tuna_kernel.py:136-139
    P_beta = clean_density_matrix(P_spin, S, n_beta)
    P = P_alpha + P_beta
    return P, P_alpha, P_beta, float(np.einsum("mn,mn->", H_core, P, optimize=True))
