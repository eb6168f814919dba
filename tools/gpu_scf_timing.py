import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
from oracle import scf_oracle as so
eng = Engine(0)
for name, sym, R, basis, nocc in [("N2/cc-pVTZ", ["N","N"], mol.angstrom_to_bohr(1.0977), "cc-pVTZ", 7), ("Ar2/cc-pVQZ", ["AR","AR"], mol.angstrom_to_bohr(3.76), "cc-pVQZ", 18)]:
    atoms = mol.make_atoms(sym, R); sh = mol.build_shells(atoms, basis); aos = mol.expand_cartesian_aos(sh)
    for rep in range(3):
        t0 = time.perf_counter(); eng.set_basis(aos); t1 = time.perf_counter(); eng.build_eri(True); t2 = time.perf_counter()
        xyz = [a.origin for a in atoms]; chg = [float(a.charge) for a in atoms]
        S, T, V, _, _ = eng.one_electron(xyz, chg, [0,0,0.5*atoms[1].origin[2]]); t3 = time.perf_counter()
        X, _, _ = eng.orthogonaliser(S); t4 = time.perf_counter()
        P0, E0 = so.core_guess(T, V, X, nocc); t5 = time.perf_counter()
        nao = [sum(s.n_sph for s in sh if s.atom == a) for a in range(2)]
        r = eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="none", n_atom_ao=nao); t6 = time.perf_counter()
        print(f"{name} rep{rep}: set_basis {t1-t0:.4f} eri {t2-t1:.4f} ({eng.eri_timings()['cart_kernel_s']:.4f} kern) 1e {t3-t2:.4f} ortho {t4-t3:.4f} guess(cpu) {t5-t4:.4f} scf {t6-t5:.4f} "
              f"[iters {r['n_iter']} fock {r['fock_seconds']:.4f} eig {r['eig_seconds']:.4f} wall {r['wall_seconds']:.4f}] E={r['energy']:.10f}")
