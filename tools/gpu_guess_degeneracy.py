"""GPU: the core guess of the bench's synthetic N-AO diatomic -- orbital energies around the occupation boundary and the SCF that follows,
with the exact eigensolves done block by block (default) and on the full matrix (TF_EIGH_BLOCKS=0, second process).
usage: python tools/gpu_guess_degeneracy.py [N=400]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
nocc = 18
atoms = mol.make_atoms(["AR", "AR"], 7.1)
sh = mol.build_shells(atoms, {18: mol.even_tempered_basis(*mol.synthetic_counts(N))})
aos = mol.expand_cartesian_aos(sh)
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    xyz = [a.origin for a in atoms]; chg = [float(a.charge) for a in atoms]
    S, T, V, _, _ = eng.one_electron(xyz, chg, [0, 0, 0.5 * atoms[1].origin[2]])
    X, smin, _ = eng.orthogonaliser(S)
    eps, C0 = eng.diagonalise(T + V, X)
    print("blocks:", os.environ.get("TF_EIGH_BLOCKS", "1"), "core-guess orbital energies", nocc - 4, "..", nocc + 3, ":", np.array2string(eps[nocc - 4:nocc + 4], precision=10))
    ref = np.linalg.eigvalsh(0.5 * ((X.T @ (T + V) @ X) + (X.T @ (T + V) @ X).T))
    print("max |eps - LAPACK|", np.abs(eps - ref).max(), " gap at the boundary", eps[nocc] - eps[nocc - 1])
    P0 = 2.0 * C0[:, :nocc] @ C0[:, :nocc].T
    P0 = 0.5 * (P0 + P0.T)
    nao = [sum(s.n_sph for s in sh if s.atom == a) for a in range(2)]
    from tuna_amd._lib import TunaError
    for damping in ("none", "dynamic"):
        for rep in range(2):
            t0 = time.perf_counter()
            try:
                r = eng.scf_rhf(S, T, V, P0, float(np.sum(P0 * (T + V))), nocc, mol.nuclear_repulsion(atoms), X=X, conv="tight", damping=damping, n_atom_ao=nao, max_iter=200)
            except TunaError as e:
                print(f"  scf damping={damping}: {e}")
                break
            dt = time.perf_counter() - t0
            print(f"  scf damping={damping}: {r['n_iter']} iterations, {dt * 1e3:.1f} ms, {1e3 * dt / r['n_iter']:.2f} ms/it, eigen {1e3 * r['eig_seconds'] / r['n_iter']:.2f} ms/it, E = {r['energy']:.10f}", eng.eigh_stats())
        tab = np.asarray(r["table"])
        print("   energies of the first 6 iterations:", np.array2string(tab[:6, 1], precision=6))
