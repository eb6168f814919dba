"""GPU: element-wise check of the first-quarter path (tfmp2::mo_q1_kernel) against the dense copy of the tensor: unit-vector coefficient
matrices pick single elements (mu nu|lambda sigma); prints the mismatching index tuples."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
eng = Engine(0)
counts = mol.synthetic_counts(n)
atoms = mol.make_atoms(["AR", "AR"], 7.1)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
eng.set_basis(aos).build_eri(True)
N = eng.N
E = eng.copy_eri()
I = np.eye(N)
rng = np.random.default_rng(0)
bad = 0
pairs = [(N - 1, N - 1), (N - 1, 0), (N - 1, N // 2), (N // 2, N // 3), (5, 2), (N - 2, N - 3)] + [tuple(sorted(rng.integers(0, N, 2), reverse=True)) for _ in range(6)]
for (mu, nu) in pairs:
    for l0 in range(0, N, 32):
        C3 = np.ascontiguousarray(I[:, l0:min(N, l0 + 32)])
        out = eng.ao_to_mo(np.ascontiguousarray(I[:, mu:mu + 1]), np.ascontiguousarray(I[:, nu:nu + 1]), C3, I)[0, 0]
        ref = E[mu, nu, l0:l0 + C3.shape[1], :]
        d = np.abs(out - ref)
        if d.max() > 1e-11:
            idx = np.argwhere(d > 1e-11)
            bad += len(idx)
            for (r, s) in idx[:6]:
                print(f"(mu,nu)=({mu},{nu}) lambda={l0 + r} sigma={s}: got {out[r, s]:.6e} ref {ref[r, s]:.6e}")
print("mismatches:", bad)
# all rows (mu, nu) for a few nu: random first ket matrix of 18 columns
C3 = rng.standard_normal((N, 18))
for nu in [0, 1, N // 3, N // 2, N - 1]:
    out = eng.ao_to_mo(I, np.ascontiguousarray(I[:, nu:nu + 1]), C3, I)[:, 0]        # [mu][r][s]
    ref = np.einsum("mls,lr->mrs", E[:, nu], C3)
    d = np.abs(out - ref).max(axis=(1, 2))
    badmu = np.where(d > 1e-10 * np.abs(ref).max())[0]
    print(f"nu={nu}: rows mu with errors: {badmu[:40].tolist()} (max err {d.max():.3e})")
    for mu in badmu[:3]:
        dd = np.abs(out[mu] - ref[mu])
        s_bad = np.where(dd.max(axis=0) > 1e-10 * np.abs(ref).max())[0]
        print("   mu", mu, "bad sigma:", s_bad[:40].tolist())
