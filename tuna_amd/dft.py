"""Kohn-Sham DFT host side: the molecular integration grid, built exactly as the reference builds
it, and the table of functionals the GPU kernels implement.  Everything per-iteration (density, functional, V_XC) runs in
libtunafock (csrc/tf_dft.hip.h).

Reference: set_up_integration_grid tuna_dft.py:94-208 (extent = mult * max(real_vdw_radius) / 6, n_radial = int(extent * acc),
Lebedev order nearest to 9 * acc), build_atomic_radial_and_angular_grid :210-258 (Gauss-Legendre in t, r = R t^3, Lebedev angular
rule from SciPy), calculate_Becke_diatomic_weights :268-322 (steepness 4, size adjustment by the vdW-radius ratio),
build_molecular_grid :332-394; grid presets tuna_util.py:129-137; functional table tuna_util.py:1440-1500.
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


def build_atomic_radial_and_angular_grid(radial_grid_cutoff, n_radial, lebedev_order, radial_power=3):     # tuna_dft.py:210-258
    from scipy.integrate import lebedev_rule
    t_nodes, t_weights = np.polynomial.legendre.leggauss(n_radial)
    t = (t_nodes + 1) / 2
    w_t = t_weights / 2
    r = radial_grid_cutoff * t ** radial_power
    dr_dt = radial_grid_cutoff * radial_power * t ** (radial_power - 1)
    weights_radial = w_t * dr_dt
    unit_sphere_directions, weights_angular = lebedev_rule(lebedev_order)
    atomic_points = np.einsum("m,in->imn", r, unit_sphere_directions, optimize=True)
    atomic_weights = np.einsum("m,m,n->mn", weights_radial, r ** 2, weights_angular, optimize=True)
    return atomic_points, atomic_weights


def calculate_Becke_diatomic_weights(X, Y, Z, bond_length, radii, steepness=4):                            # tuna_dft.py:268-322
    R_A = (X * X + Y * Y + Z * Z) ** (1 / 2)
    R_B = (X * X + Y * Y + (Z - bond_length) * (Z - bond_length)) ** (1 / 2)
    s = (R_A - R_B) / bond_length
    chi = radii[0] / radii[1]
    u = (chi - 1) / (chi + 1)
    a = u / (u * u - 1)
    s = s + a * (1 - s * s)
    for _ in range(steepness):
        s = (3 * s - s * s * s) / 2
    return (1 - s) / 2, (1 + s) / 2


def build_molecular_grid(radial_grid_cutoff, n_radial, lebedev_order, bond_length, atoms):                 # tuna_dft.py:332-394
    points_A, atomic_weights_A = build_atomic_radial_and_angular_grid(radial_grid_cutoff, n_radial, lebedev_order)
    X_A, Y_A, Z_A = points_A
    if len(atoms) == 1 or any(a.charge == 0 for a in atoms):
        return points_A, atomic_weights_A
    X = np.concatenate([X_A, X_A], axis=0)
    Y = np.concatenate([Y_A, Y_A], axis=0)
    Z = np.concatenate([Z_A, Z_A + bond_length], axis=0)
    points = np.stack((X, Y, Z), axis=0)
    wA, wB = calculate_Becke_diatomic_weights(X, Y, Z, bond_length, [real_vdw_radius(a.symbol) for a in atoms])
    n_A = X_A.shape[0]
    weights = np.concatenate([atomic_weights_A * wA[:n_A], atomic_weights_A * wB[n_A:]], axis=0)
    return points, weights


def integration_grid(atoms, grid_conv="medium"):
    """(points [3, n_r(*2), n_ang], weights [n_r(*2), n_ang], info) for the molecule (tuna_dft.py:94-160)."""
    gc = GRID_CONVERGENCE[grid_conv] if isinstance(grid_conv, str) else grid_conv
    acc, mult = gc["integral_accuracy"], gc["extent_multiplier"]
    extent = mult * np.max([real_vdw_radius(a.symbol) for a in atoms]) / 6
    n = int(acc * 9)
    lebedev_order = int(LEBEDEV_ORDERS[np.abs(LEBEDEV_ORDERS - n).argmin()])
    n_radial = int(extent * acc)
    bond_length = float(atoms[-1].origin[2] - atoms[0].origin[2]) if len(atoms) == 2 else 0.0
    points, weights = build_molecular_grid(extent, n_radial, lebedev_order, bond_length, atoms)
    return points, weights, {"n_radial": n_radial, "lebedev_order": lebedev_order, "n_angular": weights.shape[1], "extent": float(extent),
                             "n_points": int(weights.size)}
