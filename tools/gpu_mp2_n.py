"""GPU: RMP2 leg at a larger N through both paths (short index first / expanded blocks): time and agreement.  usage: python tools/gpu_mp2_n.py [N=600]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
eng = Engine(0)
counts = mol.synthetic_counts(n)
atoms = mol.make_atoms(["AR", "AR"], 7.1)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
eng.set_basis(aos).build_eri(True)
N, o = eng.N, 18
Q, _ = np.linalg.qr(np.random.default_rng(0).standard_normal((N, N)))
eps = np.concatenate([-np.arange(o, 0, -1.0), np.arange(1.0, N - o + 1)])
res = {}
for q1 in ("1", "0"):
    os.environ["TF_MO_Q1"] = q1
    for rep in range(3):
        r = eng.mp2_rhf(Q, eps, o)
    res[q1] = r
    print(f"N={N} TF_MO_Q1={q1}: {r['seconds']*1e3:.1f} ms  E_OS {r['E_OS']:.12f} E_SS {r['E_SS']:.12f}", flush=True)
print("relative difference E_OS %.2e, E_SS %.2e" % (abs(res["1"]["E_OS"] / res["0"]["E_OS"] - 1), abs(res["1"]["E_SS"] / res["0"]["E_SS"] - 1)))
