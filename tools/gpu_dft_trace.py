"""GPU: Kohn-Sham SCF trajectory of one system of tests/golden/dft_*.npz beside the reference's table (usage: python tools/gpu_dft_trace.py TAG)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import DFT_SYSTEMS
from oracle import scf_oracle as so
from tuna_amd import dft, molecule as mol
from tuna_amd.engine import Engine
tag = sys.argv[1]
g = {}
for f in ("dft_systems", "dft_functionals"):
    z = np.load(os.path.join(ROOT, "tests", "golden", f + ".npz"))
    g.update({k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(tag + "__")})
sym, R, basis, nocc, method, grid = DFT_SYSTEMS[tag]
atoms = mol.make_atoms(sym, R); shells = mol.build_shells(atoms, basis); aos = mol.expand_cartesian_aos(shells)
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    pts, wts, info = dft.integration_grid(atoms, grid)
    f = eng.dft_setup(pts, wts, method)
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    S, T, V, _, _ = eng.one_electron(xyz, chg, [0, 0, 0.0])
    X, _, _ = eng.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, nocc)
    ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
    r = eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="dynamic", hfx=f["hfx"], n_atom_ao=ranges)
ref = g["table"]
for k in range(max(r["n_iter"], len(ref))):
    a = r["table"][k] if k < r["n_iter"] else None
    b = ref[k] if k < len(ref) else None
    print(k + 1, *("%.10f damp %.6f comm %.3e" % (t[1], t[6], t[5]) if t is not None else "-" for t in (a, b)),
          "dE %.2e" % (a[1] - b[1]) if a is not None and b is not None else "")
# the first Fock matrix of the cycle (core-guess density): frontier gap -- a near-degeneracy amplifies rounding differences in P
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    f = eng.dft_setup(pts, wts, method)
    Vxc, n_el, ex, ec = eng.dft_vxc(P0)
    J, K = eng.fock_jk(P0)
    F1 = T + V + J - 0.5 * f["hfx"] * K + Vxc
    F1 = 0.5 * (F1 + F1.T)
    eps1, C1 = so.diagonalise(F1, X)
    print("first Fock matrix: eps around the Fermi level", eps1[max(0, nocc - 3):nocc + 3], "gap", eps1[nocc] - eps1[nocc - 1])
