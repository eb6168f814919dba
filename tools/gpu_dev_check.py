"""Developer smoke: HIP path vs CPU oracle on a ladder of systems (prints max abs differences)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
from oracle import oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")

def system(sym, R, basis):
    atoms = mol.make_atoms(sym, R); sh = mol.build_shells(atoms, basis)
    return atoms, sh, mol.expand_cartesian_aos(sh)

def sph(U, E):
    E = np.tensordot(U, E, axes=(1, 0))
    E = np.tensordot(U, E, axes=(1, 1)).transpose(1, 0, 2, 3)
    E = np.tensordot(U, E, axes=(1, 2)).transpose(1, 2, 0, 3)
    E = np.tensordot(U, E, axes=(1, 3)).transpose(1, 2, 3, 0)
    return np.ascontiguousarray(E)

eng = Engine(0)
R_N2 = mol.angstrom_to_bohr(1.0977)
hb = {7: [("S", [(1.3, 1.0)]), ("P", [(0.9, 1.0)]), ("D", [(1.1, 1.0)]), ("F", [(0.8, 1.0)]), ("G", [(1.0, 1.0)]), ("H", [(0.7, 1.0)])],
      8: [("S", [(2.0, 0.6), (0.5, 0.5)]), ("D", [(0.9, 0.7), (0.4, 0.4)]), ("H", [(1.2, 1.0)])]}
for tag, sym, R, basis in [("H2/STO-3G", ["H", "H"], 1.4, "STO-3G"), ("N2/STO-3G", ["N", "N"], R_N2, "STO-3G"),
                           ("He/6-31G", ["HE"], None, "6-31G"),
                           ("N2/cc-pVDZ", ["N", "N"], R_N2, "cc-pVDZ"), ("N2/cc-pVTZ", ["N", "N"], R_N2, "cc-pVTZ"),
                           ("highL", ["N", "O"], 2.1, hb)]:
    atoms, sh, aos = system(sym, R, basis)
    eng.set_basis(aos)
    xyz = [a.origin for a in atoms]; chg = [float(a.charge) for a in atoms]; org = [0, 0, 0.5 * atoms[-1].origin[2]]
    t = time.time(); Eo = orc.eri(aos); t_or = time.time() - t
    t = time.time(); eng.build_eri(False); t_g = time.time() - t
    Eg = eng.copy_eri()
    print(f"{tag}: n_cart={aos.n} ERI cart maxabs diff {np.abs(Eg-Eo).max():.3e} (max {np.abs(Eo).max():.3f}) oracle {t_or:.2f}s gpu {t_g:.3f}s", eng.eri_timings())
    U = eng.sph_matrix()
    Es = sph(U, Eo)
    t = time.time(); eng.build_eri(True); t_g = time.time() - t
    Egs = eng.copy_eri()
    print(f"   sph N={eng.N} maxabs diff {np.abs(Egs-Es).max():.3e} gpu {t_g:.3f}s storage {eng.eri_storage()}")
    rng = np.random.default_rng(0); A = rng.standard_normal((eng.N, eng.N)); P = A + A.T
    J, K = eng.fock_jk(P)
    Jr = np.einsum("ijkl,kl->ij", Es, P, optimize=True); Kr = np.einsum("ilkj,kl->ij", Es, P, optimize=True)
    print(f"   J diff {np.abs(J-Jr).max():.3e} (max {np.abs(Jr).max():.2f})  K diff {np.abs(K-Kr).max():.3e}")
    o1 = orc.one_electron(aos, xyz, chg, org)
    g1 = eng.one_electron(xyz, chg, org, spherical=False)
    print("   1e cart diffs", [f"{np.abs(a-b).max():.2e}" for a, b in zip(g1, o1)])
    nrm, cf = eng.norms(); on, oc = orc.normalize(aos)
    print("   norm diffs", np.abs(nrm - on).max(), np.abs(cf - oc).max())

# SCF vs golden (reference tuna_scf.py trajectory), N2/cc-pVTZ
g = np.load(os.path.join(GOLD, "c2_n2_ccpvtz.npz"))
atoms, sh, aos = system(["N", "N"], R_N2, "cc-pVTZ")
eng.set_basis(aos).build_eri(True)
xyz = [a.origin for a in atoms]; chg = [float(a.charge) for a in atoms]
S, T, V, D, Q = eng.one_electron(xyz, chg, [0, 0, 0.5 * atoms[1].origin[2]], spherical=True)
X, sm, Si = eng.orthogonaliser(S)
F0 = X.T @ (T + V) @ X; F0 = 0.5 * (F0 + F0.T)
w, v = np.linalg.eigh(F0); C0 = X @ v; P0 = 2 * C0[:, :7] @ C0[:, :7].T; P0 = 0.5 * (P0 + P0.T); E0 = float(np.sum(P0 * (T + V)))
nao = [sum(s.n_sph for s in sh if s.atom == a) for a in range(2)]
for damp, key in (("dynamic", ""), ("none", "_nodamp")):
    r = eng.scf_rhf(S, T, V, P0, E0, 7, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping=damp, n_atom_ao=nao)
    print(f"SCF N2/cc-pVTZ damping={damp}: E={r['energy']:.10f} golden {float(g['scf_energy'+key]):.10f} diff {r['energy']-float(g['scf_energy'+key]):.2e} "
          f"iters {r['n_iter']} (golden {len(g['scf_table'+key])}) wall {r['wall_seconds']:.3f}s fock {r['fock_seconds']:.4f}s eig {r['eig_seconds']:.3f}s")
    if key:
        n = min(len(r["table"]), len(g["scf_table" + key]))
        print("   per-iteration |dE_total| vs golden:", np.abs(r["table"][:n, 1] - g["scf_table" + key][:n, 1]).max())
