import os
import sys

import numpy as np
import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")       # as tuna_amd/__init__.py sets it: before anything initialises HIP


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built library (the .so is git-ignored): compile it once, as __graft_entry__.build() does
    try:
        from tuna_amd import _lib
        if not os.path.exists(_lib.LIB_PATH):
            _lib.build_library()
    except Exception as e:            # (hipcc missing: the ABI test reports it)
        print(f"conftest: libtunafock.so not built: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLD, name + ".npz"))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def small(golden):
    """tests/golden/small_systems.npz split by system tag."""
    z = golden("small_systems")
    out = {}
    for key in z.files:
        tag, name = key.split("__", 1)
        out.setdefault(tag, {})[name] = z[key]
    return out


@pytest.fixture(scope="session")
def engine():
    """One libtunafock context on cuda:0 -- fails loudly (no fallback) if the HIP library or the GPU is missing."""
    from tuna_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


from tuna_amd import molecule as mol  # noqa: E402

R_H2 = mol.angstrom_to_bohr(0.74)
R_N2 = mol.angstrom_to_bohr(1.0977)
R_CO = mol.angstrom_to_bohr(1.128)
R_AR2 = mol.angstrom_to_bohr(3.76)
HIGH_L_BASIS = {7: [("S", [(1.3, 1.0)]), ("P", [(0.9, 1.0)]), ("D", [(1.1, 1.0)]), ("F", [(0.8, 1.0)]), ("G", [(1.0, 1.0)]),
                    ("H", [(0.7, 1.0)])],
                8: [("S", [(2.0, 0.6), (0.5, 0.5)]), ("D", [(0.9, 0.7), (0.4, 0.4)]), ("H", [(1.2, 1.0)])]}
SYSTEMS = {
    "h2_sto3g_1p4": (["H", "H"], 1.4, "STO-3G", 1),
    "h2_sto3g": (["H", "H"], R_H2, "STO-3G", 1),
    "n2_sto3g": (["N", "N"], R_N2, "STO-3G", 7),
    "he_631g": (["HE"], None, "6-31G", 1),
    "n2_ccpvdz": (["N", "N"], R_N2, "cc-pVDZ", 7),
    "c2_n2_ccpvtz": (["N", "N"], R_N2, "cc-pVTZ", 7),
    "c4_co_def2tzvp": (["C", "O"], R_CO, "def2-TZVP", 7),
    "c3_ar2_ccpvqz": (["AR", "AR"], R_AR2, "cc-pVQZ", 18),
    "high_l": (["N", "O"], 2.1, HIGH_L_BASIS, 7),
}


def make_system(tag):
    sym, R, basis, nocc = SYSTEMS[tag]
    atoms = mol.make_atoms(sym, R)
    shells = mol.build_shells(atoms, basis)
    return atoms, shells, mol.expand_cartesian_aos(shells), nocc


def atom_arrays(atoms):
    xyz = [a.origin for a in atoms]
    chg = [float(a.charge) for a in atoms]
    org = [0.0, 0.0, 0.5 * atoms[-1].origin[2] if len(atoms) == 2 else 0.0]
    return xyz, chg, org

UHF_SYSTEMS = {
    "o2_triplet_sto3g": (["O", "O"], mol.angstrom_to_bohr(1.2075), "STO-3G", 9, 7),
    "o2_triplet_ccpvdz": (["O", "O"], mol.angstrom_to_bohr(1.2075), "cc-pVDZ", 9, 7),
    "no_doublet_631g": (["N", "O"], mol.angstrom_to_bohr(1.1508), "6-31G", 8, 7),
    "oh_doublet_ccpvdz": (["O", "H"], mol.angstrom_to_bohr(0.9697), "cc-pVDZ", 5, 4),
    "li_doublet_631g": (["LI"], None, "6-31G", 2, 1),
}


def make_uhf_system(tag):
    sym, R, basis, na, nb = UHF_SYSTEMS[tag]
    atoms = mol.make_atoms(sym, R)
    shells = mol.build_shells(atoms, basis)
    return atoms, shells, mol.expand_cartesian_aos(shells), na, nb


@pytest.fixture(scope="session")
def uhf_golden(golden):
    z = golden("uhf_systems")
    out = {}
    for key in z.files:
        tag, name = key.split("__", 1)
        out.setdefault(tag, {})[name] = z[key]
    return out


@pytest.fixture(scope="session")
def mp2_golden(golden):
    z = golden("mp2_systems")
    out = {}
    for key in z.files:
        tag, name = key.split("__", 1)
        out.setdefault(tag, {})[name] = z[key]
    return out


MP2_SYSTEMS = {"n2_sto3g": "n2_sto3g", "n2_ccpvdz": "n2_ccpvdz", "c5_n2_ccpvtz": "c2_n2_ccpvtz", "co_631g": None}


DFT_SYSTEMS = {
    "h2_lda_sto3g": (["H", "H"], R_H2, "STO-3G", 1, "LDA", "loose"),
    "n2_blyp_631g": (["N", "N"], R_N2, "6-31G", 7, "BLYP", "loose"),
    "co_b3lyp_631g": (["C", "O"], R_CO, "6-31G", 7, "B3LYP", "medium"),
    "co_b3lypg_ccpvdz": (["C", "O"], R_CO, "cc-pVDZ", 7, "B3LYP/G", "loose"),
    "c4_co_b3lyp_def2tzvp": (["C", "O"], R_CO, "def2-TZVP", 7, "B3LYP", "medium"),
    # one small system per remaining functional of tuna_amd.dft.FUNCTIONALS (tests/golden/dft_functionals.npz, --dft-sweep-only)
    "lih_hfs_sto3g": (["LI", "H"], mol.angstrom_to_bohr(1.595), "STO-3G", 2, "HFS", "loose"),
    "lih_svwn3_sto3g": (["LI", "H"], mol.angstrom_to_bohr(1.595), "STO-3G", 2, "SVWN3", "loose"),
    "hf_hfb_631g": (["F", "H"], mol.angstrom_to_bohr(0.917), "6-31G", 5, "HFB", "loose"),
    "hf_bvwn_631g": (["F", "H"], mol.angstrom_to_bohr(0.917), "6-31G", 5, "BVWN", "loose"),
    "lih_bvwn3_sto3g": (["LI", "H"], mol.angstrom_to_bohr(1.595), "STO-3G", 2, "BVWN3", "loose"),
    "hf_bhlyp_631g": (["F", "H"], mol.angstrom_to_bohr(0.917), "6-31G", 5, "BHLYP", "loose"),
    "hf_b1lyp_631g": (["F", "H"], mol.angstrom_to_bohr(0.917), "6-31G", 5, "B1LYP", "loose"),
    "lih_slyp_sto3g": (["LI", "H"], mol.angstrom_to_bohr(1.595), "STO-3G", 2, "SLYP", "loose"),
}


@pytest.fixture(scope="session")
def dft_golden(golden):
    out = {}
    for name_ in ("dft_systems", "dft_functionals"):
        z = golden(name_)
        for key in z.files:
            tag, name = key.split("__", 1)
            out.setdefault(tag, {})[name] = z[key]
    return out
