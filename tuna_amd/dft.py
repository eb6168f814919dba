"""Kohn-Sham DFT host side: the molecular integration grid, built exactly as the reference builds
it, and the table of functionals the GPU kernels implement.  Everything per-iteration (density, functional, V_XC) runs in
libtunafock (csrc/tf_dft.hip.h).

Reference: set_up_integration_grid tuna_dft.py:94-208 (extent = mult * max(real_vdw_radius) / 6, n_radial = int(extent * acc),
Lebedev order nearest to 9 * acc), the atomic product grid :210-258 (Gauss-Legendre in u, r = R u^3, Lebedev angular rule from
SciPy) -> atomic_grid, the Becke diatomic weights :268-322 (four smoothing steps, size adjustment by the vdW-radius ratio) ->
becke_cell_functions, the molecular grid :332-394 -> molecular_grid; grid presets tuna_util.py:129-137; functional table tuna_util.py:1440-1500.
"""
from __future__ import annotations

import numpy as np

from .guess import atomic_data

GRID_CONVERGENCE = {                                    # tuna_util.py:129-137
    "loose": {"integral_accuracy": 3, "extent_multiplier": 0.7, "name": "loose"},
    "medium": {"integral_accuracy": 4, "extent_multiplier": 0.9, "name": "medium"},
    "tight": {"integral_accuracy": 5, "extent_multiplier": 1, "name": "tight"},
    "extreme": {"integral_accuracy": 7, "extent_multiplier": 1.3, "name": "extreme"},
}
LEBEDEV_ORDERS = np.array([3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 25, 27, 29, 31, 35, 41, 47, 53, 59, 65, 71, 77, 83, 89, 95, 101, 107, 113,
                           119, 125, 131])

X_ID = {None: 0, "S": 1, "B": 2, "B3": 3}
C_ID = {None: 0, "VWN5": 1, "VWN3": 2, "LYP": 3, "3P": 4, "3P/G": 5}
# name -> (x functional, c functional, DFX, HFX, DFC)      (tuna_util.py:1440-1475; only combinations of the kernels above)
FUNCTIONALS = {
    "HFS": ("S", None, 1.0, 0.0, 0.0), "SVWN": ("S", "VWN5", 1.0, 0.0, 1.0), "LSDA": ("S", "VWN5", 1.0, 0.0, 1.0),
    "LDA": ("S", "VWN5", 1.0, 0.0, 1.0), "SVWN5": ("S", "VWN5", 1.0, 0.0, 1.0), "SVWN3": ("S", "VWN3", 1.0, 0.0, 1.0),
    "HFB": ("B", None, 1.0, 0.0, 0.0), "BVWN": ("B", "VWN5", 1.0, 0.0, 1.0), "BVWN5": ("B", "VWN5", 1.0, 0.0, 1.0),
    "BVWN3": ("B", "VWN3", 1.0, 0.0, 1.0), "BLYP": ("B", "LYP", 1.0, 0.0, 1.0), "BHLYP": ("B", "LYP", 0.50, 0.50, 1.0),
    "B1LYP": ("B", "LYP", 0.75, 0.25, 1.0), "SLYP": ("S", "LYP", 1.0, 0.0, 1.0), "B3LYP": ("B3", "3P", 0.80, 0.20, 1.0),
    "B3LYP/G": ("B3", "3P/G", 0.80, 0.20, 1.0),
}


def real_vdw_radius(symbol: str) -> float:
    s = symbol.upper()
    s = s[1:] if s.startswith("X") and s[1:] in atomic_data() else s
    return float(atomic_data()[s]["real_vdw_radius"])


def atomic_grid(r_max, n_radial, lebedev_order, power=3):
    """One atom-centred product grid: points [3, n_radial, n_ang] and weights [n_radial, n_ang].

    Radial part: Gauss-Legendre nodes x on [-1, 1] mapped to u = (x + 1) / 2 in [0, 1] and then to r = r_max * u**power, so the
    volume element r^2 dr carries the Jacobian r_max * power * u**(power - 1) / 2 per Legendre weight.  Angular part: SciPy's
    Lebedev rule (its weights already sum to 4 pi).  Same quadrature as the reference builds (tuna_dft.py:210-258).
    """
    from scipy.integrate import lebedev_rule
    x, wx = np.polynomial.legendre.leggauss(n_radial)
    u = 0.5 * (x + 1.0)
    radius = r_max * u ** power
    shell_weight = (0.5 * wx) * (r_max * power * u ** (power - 1)) * radius * radius          # w_r * r^2 * dr/du
    directions, w_ang = lebedev_rule(lebedev_order)                                          # [3, n_ang], [n_ang]
    return directions[:, None, :] * radius[None, :, None], shell_weight[:, None] * w_ang[None, :]


def becke_cell_functions(points, bond_length, radii, n_smooth=4):
    """Becke's fuzzy-cell weights (w_A, w_B) of a diatomic A at the origin, B at (0, 0, bond_length); tuna_dft.py:268-322.

    mu = (|r - A| - |r - B|) / R is shifted by the atomic-size correction nu = mu + a (1 - mu^2) with a = u / (u^2 - 1),
    u = (chi - 1) / (chi + 1), chi = radius_A / radius_B, and then smoothed n_smooth times with p(t) = (3 t - t^3) / 2.
    """
    x, y, z = points
    rho2 = x * x + y * y
    mu = (np.sqrt(rho2 + z * z) - np.sqrt(rho2 + (z - bond_length) ** 2)) / bond_length
    chi = radii[0] / radii[1]
    u = (chi - 1.0) / (chi + 1.0)
    nu = mu + (u / (u * u - 1.0)) * (1.0 - mu * mu)
    for _ in range(n_smooth):
        nu = (3.0 * nu - nu * nu * nu) / 2.0         # Becke's p(t); this association keeps the near-zero cell weights bit-identical
    return 0.5 * (1.0 - nu), 0.5 * (1.0 + nu)


def molecular_grid(r_max, n_radial, lebedev_order, bond_length, atoms):
    """Atom A's grid followed by the same grid shifted to atom B, each weighted by its Becke cell function (tuna_dft.py:332-394);
    a single atom (or an atom next to a ghost centre) keeps the bare atomic grid."""
    pts, w = atomic_grid(r_max, n_radial, lebedev_order)
    if len(atoms) == 1 or any(a.charge == 0 for a in atoms):
        return pts, w
    shifted = pts + np.array([0.0, 0.0, bond_length])[:, None, None]
    both = np.concatenate([pts, shifted], axis=1)
    cell_A, cell_B = becke_cell_functions(both, bond_length, [real_vdw_radius(a.symbol) for a in atoms])
    n = w.shape[0]
    return both, np.concatenate([w * cell_A[:n], w * cell_B[n:]], axis=0)


def integration_grid(atoms, grid_conv="medium"):
    """(points [3, n_r(*2), n_ang], weights [n_r(*2), n_ang], info) for the molecule (tuna_dft.py:94-160)."""
    gc = GRID_CONVERGENCE[grid_conv] if isinstance(grid_conv, str) else grid_conv
    acc, mult = gc["integral_accuracy"], gc["extent_multiplier"]
    extent = mult * np.max([real_vdw_radius(a.symbol) for a in atoms]) / 6
    n = int(acc * 9)
    lebedev_order = int(LEBEDEV_ORDERS[np.abs(LEBEDEV_ORDERS - n).argmin()])
    n_radial = int(extent * acc)
    bond_length = float(atoms[-1].origin[2] - atoms[0].origin[2]) if len(atoms) == 2 else 0.0
    points, weights = molecular_grid(extent, n_radial, lebedev_order, bond_length, atoms)
    return points, weights, {"n_radial": n_radial, "lebedev_order": lebedev_order, "n_angular": weights.shape[1], "extent": float(extent),
                             "n_points": int(weights.size)}
