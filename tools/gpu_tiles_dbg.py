"""GPU: reproduce tests/test_gpu_tiles.py::test_tiles_tensor_and_fock_matrices_against_reference_golden step by step (debugging aid)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
R_N2 = mol.angstrom_to_bohr(1.0977)
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "n2_ccpvdz.npz"))
atoms = mol.make_atoms(["N", "N"], R_N2)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, "cc-pVDZ"))
eng = Engine(0)
eng.set_basis(aos)
eng.build_eri(False, layout=sys.argv[1] if len(sys.argv) > 1 else "tiles")
import zlib
def zt(tag):
    zlib.decompressobj(-15); print("zlib ok after", tag, flush=True)
zt("build")
print("built", eng.eri_storage(), flush=True)
idx, val = z["eri_idx"].astype(np.int32), z["eri_val"]
print("idx", idx.shape, idx.dtype, val.shape, flush=True)
got = eng.sample_eri(idx)
print("sampled", np.abs(got - val).max(), flush=True); zt("sample")
P = z["P_rand"]
print("P sym", np.abs(P - P.T).max(), flush=True)
J, K = eng.fock_jk(P)
print("jk", flush=True); zt("jk"); import gc; gc.collect(); zt("gc")
Jr = z["J_rand"]
print("loaded J", np.abs(J - Jr).max() / np.abs(Jr).max(), np.abs(K - z["K_rand"]).max() / np.abs(z["K_rand"]).max(), flush=True)
