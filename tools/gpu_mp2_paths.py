import os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
eng = Engine(0)
counts = mol.synthetic_counts(120)
atoms = mol.make_atoms(["AR", "AR"], 7.1)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
eng.set_basis(aos).build_eri(True)
N = eng.N
rng = np.random.default_rng(3)
C1, C2, C3, C4 = rng.standard_normal((N, 4)), rng.standard_normal((N, 6)), rng.standard_normal((N, 18)), rng.standard_normal((N, 7))
os.environ["TF_MO_Q1"] = "1"; a = eng.ao_to_mo(C1, C2, C3, C4)
os.environ["TF_MO_Q1"] = "0"; b = eng.ao_to_mo(C1, C2, C3, C4)
print("max |fast - blocks| =", np.abs(a - b).max(), "scale", np.abs(a).max(), "identical:", np.array_equal(a, b))
Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
eps = np.concatenate([-np.arange(18, 0, -1.0), np.arange(1.0, N - 18 + 1)])
os.environ["TF_MO_Q1"] = "1"; r1 = eng.mp2_rhf(Q, eps, 18)
os.environ["TF_MO_Q1"] = "0"; r0 = eng.mp2_rhf(Q, eps, 18)
print("E_OS", repr(r1["E_OS"]), repr(r0["E_OS"]), "E_SS", repr(r1["E_SS"]), repr(r0["E_SS"]), r1["seconds"], r0["seconds"])
