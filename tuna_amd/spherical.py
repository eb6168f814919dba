"""Cartesian -> real spherical-harmonic AO transformation, derived from the closed formula.

The reference stores these blocks as literal tables (tuna_kernel.py:554-623) and assembles the
block-diagonal U of shape (n_sph, n_cart) shell by shell (tuna_kernel.py:629-649).  Here the
coefficients come from the closed form for real solid harmonics expressed in *individually
normalised* Cartesian Gaussians (Schlegel & Frisch, Int. J. Quantum Chem. 54, 83 (1995), eq. 15),
with the reference's conventions:
  * Cartesian order x^L ... z^L of tuna_molecule.py:622,
  * rows ordered m = -L..+L, no Condon-Shortley phase,
  * except P, which stays (x, y, z), and D, whose rows are (xy, xz, yz, x2-y2, z2)
    (tuna_kernel.py:556-568).
tests/test_spherical.py checks every block against the reference's tables (tests/golden/sph_blocks.npz).
"""
from __future__ import annotations

from functools import lru_cache
from math import comb, factorial, sqrt

import numpy as np


def _cart_components(L):
    return [(i, j, L - i - j) for i in range(L, -1, -1) for j in range(L - i, -1, -1)]


def _complex_coeff(L, m, lx, ly, lz):
    """Coefficient of normalised x^lx y^ly z^lz in the complex solid harmonic Y_L^{|m|} (Schlegel-Frisch)."""
    am = abs(m)
    j2 = lx + ly - am
    if j2 < 0 or j2 % 2:
        return 0.0 + 0.0j
    j = j2 // 2
    pref = sqrt(factorial(2 * lx) * factorial(2 * ly) * factorial(2 * lz) * factorial(L) * factorial(L - am)
                / (factorial(2 * L) * factorial(lx) * factorial(ly) * factorial(lz) * factorial(L + am)))
    pref /= (2 ** L) * factorial(L)
    s1 = 0.0
    for i in range((L - am) // 2 + 1):
        if j > i:
            continue
        s1 += comb(L, i) * comb(i, j) * (-1) ** i * factorial(2 * L - 2 * i) / factorial(L - am - 2 * i)
    s2 = 0.0 + 0.0j
    for k in range(j + 1):
        if 0 <= lx - 2 * k <= am:
            s2 += comb(j, k) * comb(am, lx - 2 * k) * (1j) ** (am - lx + 2 * k)
    return pref * s1 * s2


@lru_cache(maxsize=None)
def spherical_block(L: int) -> np.ndarray:
    """(2L+1, (L+1)(L+2)/2) block mapping normalised Cartesians to the reference's real harmonics."""
    comps = _cart_components(L)
    if L == 0:
        return np.eye(1)
    if L == 1:
        return np.eye(3)
    rows = {}
    for m in range(-L, L + 1):
        row = np.zeros(len(comps))
        for c, (lx, ly, lz) in enumerate(comps):
            z = _complex_coeff(L, m, lx, ly, lz)
            if m == 0:
                row[c] = z.real
            elif m > 0:
                row[c] = sqrt(2.0) * z.real
            else:
                row[c] = sqrt(2.0) * z.imag
        rows[m] = row
    order = [-2, 1, -1, 2, 0] if L == 2 else list(range(-L, L + 1))
    return np.array([rows[m] for m in order])


def transformation_matrix(shell_L: list[int]) -> np.ndarray:
    """Block-diagonal U (n_sph x n_cart) over shells in AO order (tuna_kernel.py:629-649)."""
    n_s = sum(2 * L + 1 for L in shell_L)
    n_c = sum((L + 1) * (L + 2) // 2 for L in shell_L)
    U = np.zeros((n_s, n_c))
    r = c = 0
    for L in shell_L:
        B = spherical_block(L)
        U[r:r + B.shape[0], c:c + B.shape[1]] = B
        r += B.shape[0]
        c += B.shape[1]
    return U
