"""RMP2 / AO->MO timing on the synthetic series (random orthonormal orbitals; flops counted for the ovov transformation)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
eng = Engine(0)
for n in (int(a) for a in (sys.argv[1:] or ["120", "200", "300", "400"])):
    counts = mol.synthetic_counts(n)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
    eng.set_basis(aos).build_eri(True)
    N, o = eng.N, 18
    v = N - o
    Q, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((N, N)))
    eps = np.concatenate([-np.arange(o, 0, -1.0), np.arange(1.0, v + 1)])
    r = eng.mp2_rhf(Q, eps, o)            # warm-up (rocBLAS kernels)
    r = eng.mp2_rhf(Q, eps, o)
    rows = N * (N + 1) // 2
    flops = rows * (2.0 * o * N * N + 2.0 * o * N * v) + 2.0 * o * N * N * o * v + o * 2.0 * v * N * o * v
    print(f"N={N}: RMP2 (ovov transform + energy) {r['seconds']*1e3:.1f} ms, {flops/r['seconds']/1e12:.2f} TFLOP/s f64, "
          f"tensor pass {eng.eri_storage()['bytes']/r['seconds']/1e12:.2f} TB/s equivalent, E_MP2(random orbitals)={r['E_MP2']:.6f}")
