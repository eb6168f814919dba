import sys, os, time
import numpy as np
sys.path.insert(0, "/root/repo")
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
eng = Engine(0)
counts = mol.synthetic_counts(400)
atoms = mol.make_atoms(["AR", "AR"], 7.1)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
eng.set_basis(aos).build_eri(True)
N, o = eng.N, 18
Q, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((N, N)))
eps = np.concatenate([-np.arange(o, 0, -1.0), np.arange(1.0, N - o + 1)])
for rep in range(6):
    t0=time.perf_counter(); r = eng.mp2_rhf(Q, eps, o); dt=time.perf_counter()-t0
    print(f"rep {rep}: lib seconds {r['seconds']*1e3:.1f} ms, wall {dt*1e3:.1f} ms", flush=True)
