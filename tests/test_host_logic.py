"""CPU: host-side logic of the product (AO ordering, spherical coefficients, basis lookup) and the C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, make_system
from tuna_amd import _lib, molecule as mol, spherical


def test_spherical_blocks_match_reference_tables(golden):
    g = golden("sph_blocks")
    for L in range(6):
        np.testing.assert_allclose(spherical.spherical_block(L), g[f"L{L}"], atol=3e-16, rtol=0)


@pytest.mark.parametrize("tag", ["n2_ccpvdz", "c2_n2_ccpvtz", "c4_co_def2tzvp", "c3_ar2_ccpvqz", "high_l"])
def test_ao_order_and_U_match_reference(golden, tag):
    g = golden(tag)
    atoms, shells, aos, _ = make_system(tag)
    np.testing.assert_array_equal(aos.lmn, g["lmn"])
    np.testing.assert_array_equal(aos.prim_off, g["prim_off"]) if "prim_off" in g.files else None
    U = spherical.transformation_matrix([s.L for s in shells])
    np.testing.assert_allclose(U, g["U"], atol=3e-16)


def test_dimensions_of_baseline_configs():
    dims = {"c2_n2_ccpvtz": (70, 60), "c3_ar2_ccpvqz": (148, 118), "c4_co_def2tzvp": (72, 62), "h2_sto3g": (2, 2)}
    for tag, (nc, ns) in dims.items():
        _, shells, aos, _ = make_system(tag)
        assert aos.n == nc and sum(s.n_sph for s in shells) == ns


def test_units_and_names():
    assert abs(mol.BOHR_RADIUS_IN_ANGSTROM - 0.5291772105443463) < 1e-15
    assert abs(mol.angstrom_to_bohr(0.74) - 1.398397333170844) < 1e-13
    assert mol.mangle_basis_name("cc-pVTZ") == "CC_PVTZ"
    assert mol.mangle_basis_name("6-31G*") == "_6_31GSTAR"
    assert mol.mangle_basis_name("def2-TZVP") == "DEF2_TZVP"
    with pytest.raises(KeyError):
        mol.atomic_basis("no-such-basis", 1)
    assert mol.cartesian_components(2) == [(2, 0, 0), (1, 1, 0), (1, 0, 1), (0, 2, 0), (0, 1, 1), (0, 0, 2)]


def test_synthetic_series_sizes():
    b = mol.even_tempered_basis(20, 15, 13, 10)
    atoms = mol.make_atoms(["AR", "AR"], 7.1)
    shells = mol.build_shells(atoms, {18: b})
    assert sum(s.n_sph for s in shells) == 400 and sum(s.n_cart for s in shells) == 486
    for n in (100, 200, 300, 400):
        c = mol.synthetic_counts(n)
        assert 2 * (c[0] + 3 * c[1] + 5 * c[2] + 7 * c[3]) == n


def test_ghost_atoms_and_nuclear_repulsion():
    atoms = mol.make_atoms(["XH", "H"], 1.4)
    assert atoms[0].charge == 0 and atoms[0].Z == 1
    assert mol.nuclear_repulsion(atoms) == 0.0
    assert abs(mol.nuclear_repulsion(mol.make_atoms(["N", "N"], 2.0)) - 24.5) < 1e-14


# ---- C ABI -----------------------------------------------------------------------------------------

def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tunafock.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z_0-9]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/tunafock.h but not exported"
    assert sorted(_lib.EXPORTS) == declared


def test_tf_normalize_matches_reference_norms(golden):
    """tf_normalize is host-only (no GPU needed): Basis.normalize, pyx:174-210."""
    g = golden("c3_ar2_ccpvqz")
    _, _, aos, _ = make_system("c3_ar2_ccpvqz")
    L = _lib.lib()
    for i in range(aos.n):
        a, b = int(aos.prim_off[i]), int(aos.prim_off[i + 1])
        e = np.ascontiguousarray(aos.exps[a:b]); c = np.ascontiguousarray(aos.coefs[a:b]); nrm = np.zeros(b - a)
        assert L.tf_normalize(int(aos.lmn[i, 0]), int(aos.lmn[i, 1]), int(aos.lmn[i, 2]), b - a, _lib.ptr(e), _lib.ptr(c), _lib.ptr(nrm)) == 0
        np.testing.assert_allclose(nrm, g["norm"][a:b], rtol=1e-15)
        np.testing.assert_allclose(c, g["coefs"][a:b], rtol=1e-15)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tuna_amd.engine import Engine
    with pytest.raises(_lib.TunaError) as e:
        Engine(0)
    assert "no CPU fallback" in str(e.value)


def test_shard_plan_is_balanced_and_complete():
    L = _lib.lib()
    rng = np.random.default_rng(0)
    w = rng.integers(1, 400, size=595).astype(np.int64)
    for world in (1, 2, 4, 8):
        owner = np.full(len(w), -1, dtype=np.int32)
        assert L.tf_shard_plan(len(w), _lib.ptr(w), world, _lib.ptr(owner)) == 0
        assert owner.min() == 0 and owner.max() == world - 1
        loads = np.array([w[owner == r].sum() for r in range(world)])
        assert loads.sum() == w.sum() and (loads.max() - loads.min()) <= w.max()


def test_sad_guess_host_algebra(golden):
    """Projection + trace cleaning of the SAD guess (tuna_guess.py:209-236, tuna_dft.py:35-41) with oracle integrals against the
    density the reference's own functions produced (tests/golden/sad_default_runs.npz)."""
    from oracle import oracle as orc
    from oracle import scf_oracle as so
    from tuna_amd import guess
    z = golden("sad_default_runs")
    atoms, shells, aos, nocc = make_system("n2_ccpvdz")
    U = spherical.transformation_matrix([s.L for s in shells])
    xyz = [a.origin for a in atoms]; chg = [float(a.charge) for a in atoms]
    S, T, V, _, _ = orc.one_electron(aos, xyz, chg, [0, 0, 1.0])
    S, T, V = (so.to_spherical(U, M) for M in (S, T, V))
    _, _, S_inv = so.orthogonaliser(S)

    class FakeEngine:
        def cross_overlap(self, other):
            return orc.cross_overlap(aos, other)
    P, Pa, Pb, E0 = guess.superposition_guess(FakeEngine(), atoms, S, S_inv, U, nocc, nocc, T + V)
    assert np.abs(P - z["n2_ccpvdz__P_guess"]).max() < 1e-10
    assert abs(E0 - float(z["n2_ccpvdz__E_guess"])) < 1e-8
    assert abs(np.trace(Pa @ S) - nocc) < 1e-12
    assert abs(guess.centre_of_mass(mol.make_atoms(["C", "O"], 2.0)) - 2.0 * 15.994915 / (12.0 + 15.994915)) < 1e-12


@pytest.mark.parametrize("tag", ["h2_lda_sto3g", "co_b3lyp_631g", "c4_co_b3lyp_def2tzvp"])
def test_integration_grid_matches_reference(dft_golden, tag):
    """Gauss-Legendre x Lebedev atomic grids with Becke diatomic weights (tuna_dft.py:94-394) against the reference's grid."""
    from conftest import DFT_SYSTEMS
    from tuna_amd import dft
    g = dft_golden[tag]
    sym, R, basis, nocc, method, grid = DFT_SYSTEMS[tag]
    atoms = mol.make_atoms(sym, R)
    pts, wts, info = dft.integration_grid(atoms, grid)
    assert info["n_points"] == int(g["n_points"]) and info["n_radial"] == int(g["n_radial"]) and info["lebedev_order"] == int(g["lebedev"])
    assert abs(wts.sum() - float(g["weights_sum"])) < 1e-9 * abs(float(g["weights_sum"]))
    pick = g["pick"]
    np.testing.assert_allclose(pts.reshape(3, -1)[:, pick], g["pts_pick"], atol=1e-13)
    np.testing.assert_allclose(wts.reshape(-1)[pick], g["w_pick"], rtol=1e-12, atol=1e-300)
    if tag.startswith("c4"):
        assert info["n_points"] == 88536


def test_packed_index_mirror_matches_the_library():
    """tuna_amd.distributed.packed_tri_offset / packed_row_length against the layout constants of the library (tf_packed_pad):
    triangle rows and tensor rows start at multiples of the alignment unit, one 128-byte cache line."""
    from tuna_amd import distributed as tdist
    pad = tdist.packed_pad()
    assert pad == 16
    k = np.arange(0, 300)
    off = tdist.packed_tri_offset(k)
    assert off[0] == 0 and np.all(off % pad == 0)
    assert np.all(np.diff(off) == ((k[:-1] + 1 + pad - 1) // pad) * pad)          # row k holds k + 1 pairs, rounded up to the unit
    i, j = 37, 11
    assert tdist.packed_row_length(i, j) == ((off[i] + j + 1 + pad - 1) // pad) * pad
    assert np.array_equal(tdist.packed_tri_offset(k, 2), 2 * (k >> 1) * ((k >> 1) + 1) + np.where(k & 1, k + 1, 0))


def test_host_blas_pool_is_capped():
    """The package caps NumPy's BLAS pool on import (a 256-thread OpenBLAS pool spinning inside a 16-CPU quota gets the whole
    process throttled, HIP runtime threads included: tuna_amd/__init__.py)."""
    import tuna_amd
    from threadpoolctl import threadpool_info
    assert tuna_amd.cpu_quota() >= 1
    tuna_amd.limit_host_threads()
    import scipy.linalg  # noqa: F401  (SciPy's own OpenBLAS: covered whether it was loaded before or after the cap)
    tuna_amd.limit_host_threads()
    if os.environ.get("TUNA_AMD_HOST_BLAS_THREADS") is None:
        assert all(p["num_threads"] <= 4 for p in threadpool_info() if p.get("user_api") == "blas")
    tuna_amd.limit_host_threads(2)
    assert all(p["num_threads"] <= 2 for p in threadpool_info() if p.get("user_api") == "blas")
    tuna_amd.limit_host_threads()


def test_finite_difference_formulas_against_the_reference_values():
    """tuna_amd/properties.py's stencils on the reference's own field energies give the reference's derivatives
    (tests/golden/field_systems.json: tuna_util.py:581-680 evaluated from the source text)."""
    import json
    import os
    from conftest import GOLD
    from tuna_amd import properties as props
    for tag, g in json.load(open(os.path.join(GOLD, "field_systems.json"))).items():
        h1, h2, h3 = g["steps"]
        assert (h1, h2, h3) == (props.FIRST_ELEC_DERIVATIVE_STEP, props.SECOND_ELEC_DERIVATIVE_STEP, props.THIRD_ELEC_DERIVATIVE_STEP)
        d, p, y = g["dipole_energies"], g["polarisability_energies"], g["hyperpolarisability_energies"]
        assert -props.calculate_first_derivative(d["-z"], d["+z"], h1) == g["electronic_dipole"]
        assert -props.calculate_second_derivative(p["-2z"], p["-z"], g["energy"], p["+z"], p["+2z"], h2) == g["polarisability_parallel"]
        assert -props.calculate_second_derivative(p["-2x"], p["-x"], g["energy"], p["+x"], p["+2x"], h2) == g["polarisability_perpendicular"]
        assert -props.calculate_third_derivative(y["-4z"], y["-3z"], y["-2z"], y["-z"], y["+z"], y["+2z"], y["+3z"], y["+4z"], h3) == \
            g["hyperpolarisability_parallel"]


def test_lockstep_cycles_on_a_cpu_stand_in():
    """tuna_amd.scf.run_cycles_in_lockstep (the finite-field batch driver) with a stand-in for the device tensor: J/K from the
    oracle's dense tensor for a whole batch of densities, NumPy diagonalisation.  Every cycle must follow the oracle's own RHF loop
    (scf_oracle.run_rhf with the same external term) and the batch must ask for fewer Fock builds than the cycles have iterations."""
    import types
    from conftest import atom_arrays
    from oracle import oracle as orc
    from oracle import scf_oracle as so
    from tuna_amd import molecule as mol, properties as props, scf, spherical
    atoms = mol.make_atoms(["LI", "H"], mol.angstrom_to_bohr(1.595))
    shells = mol.build_shells(atoms, "STO-3G")
    aos = mol.expand_cartesian_aos(shells)
    U = spherical.transformation_matrix([s.L for s in shells])
    xyz, chg, org = atom_arrays(atoms)
    S, T, V, D, Q = orc.one_electron(aos, xyz, chg, org)
    S, T, V = (U @ M @ U.T for M in (S, T, V))
    D = np.array([U @ D[k] @ U.T for k in range(3)])
    E = so.eri_to_spherical(U, orc.eri(aos))
    X, _, _ = so.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, 2)
    ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(2)]

    class FakeEngine:
        world = 1

        def diagonalise(self, F, Xm):
            return so.diagonalise(F, Xm)

    class FakeERI:
        engine = FakeEngine()
        n_builds = 0

        def jk(self, P):
            self.n_builds += 1
            P = np.asarray(P)
            if P.ndim == 2:
                return so.coulomb(P, E), so.exchange(P, E)
            return np.array([so.coulomb(p, E) for p in P]), np.array([so.exchange(p, E) for p in P])

    eri = FakeERI()
    integrals = types.SimpleNamespace(S=S, T=T, V_NE=V, D=D, F=np.zeros_like(S), G=np.zeros_like(S), ERI_AO=eri)
    molecule = types.SimpleNamespace(atoms=atoms, n_doubly_occ=2, partition_ranges=ranges)
    from tuna_amd.engine import SCF_CONVERGENCE
    calc = types.SimpleNamespace(reference="RHF", DFT_calculation=False, SCF_conv=SCF_CONVERGENCE["tight"], max_iter=100, DIIS=True,
                                 max_DIIS_matrices=6, damping=True, damping_factor=None, max_damping=0.7, HFX_prop=1.0)
    h = props.SECOND_ELEC_DERIVATIVE_STEP
    fields = [[0, 0, 2 * h], [0, 0, h], [0, 0, -h], [0, 0, -2 * h], [h, 0, 0]]
    terms = [props.apply_electric_field(D, f) for f in fields]
    V_NN = mol.nuclear_repulsion(atoms)
    res = scf.run_cycles_in_lockstep(molecule, calc, integrals, V_NN, X, (P0, P0 / 2, P0 / 2, E0), terms)
    total_iterations = 0
    for r, term in zip(res, terms):
        ref = so.run_rhf(S, T, V, E, X, P0, E0, 2, V_NN, ranges, conv="tight", damping="dynamic", Fext=term)
        assert abs(r["energy"] - ref["energy"]) < 1e-10 and r["n_iter"] == ref["n_iter"]
        np.testing.assert_allclose(r["table"][:, 1], ref["table"][:, 1], atol=1e-10)
        total_iterations += r["n_iter"]
    assert eri.n_builds == max(r["n_iter"] for r in res) < total_iterations


def test_public_header_is_valid_c():
    """include/tunafock.h is the C ABI a maintainer binds (cgo / ctypes / JNI): it must compile as C, not only as the C++ the library is
    built from"""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    hdr = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "tunafock.h")
    r = subprocess.run([cc, "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", hdr], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_custom_criteria_are_applied_over_the_derivative_driven_preset():
    """tuna_calc.py:473-494: the named criteria are chosen first -- DIPOLE without LOOSE..EXTREME means tight, POLAR / HYPER extreme -- and
    ECONV / RMSDP / MAXDP / DIISERR then overwrite single entries of that set."""
    from tuna_amd import energy as en
    def conv(line):
        _, method, basis, _, _, params = en.parse_input(line)
        return en.interpret_keywords(params, en.Calculation("SPE", "HF", basis)).SCF_conv
    base = "SPE : H H 0.74 : HF STO-3G :"
    assert conv(base) == en.SCF_CONVERGENCE["medium"]
    c = conv(base + " DIPOLE ECONV 3e-7")
    assert c["delta_E"] == 3e-7 and {k: v for k, v in c.items() if k != "delta_E"} == {k: v for k, v in en.SCF_CONVERGENCE["tight"].items() if k != "delta_E"}
    c = conv(base + " POLAR MAXDP 1e-5 DIISERR 2e-6")
    ext = en.SCF_CONVERGENCE["extreme"]
    assert c["max_DP"] == 1e-5 and c["commutator"] == 2e-6 and c["delta_E"] == ext["delta_E"] and c["RMS_DP"] == ext["RMS_DP"]
    c = conv(base + " HYPER LOOSE RMSDP 1e-4")                        # an explicit preset wins over the derivative-driven one
    assert c["RMS_DP"] == 1e-4 and c["delta_E"] == en.SCF_CONVERGENCE["loose"]["delta_E"]


def test_bench_reads_the_committed_profiles_of_this_round():
    """bench.py turns committed rocprofv3 PMC summaries into roofline.traffic, eri_build.executed and per_density.mfma: the files it
    looks for must exist, carry `_meta`, and give sane figures (MI355X_MICROARCH.md: FETCH_SIZE x 2 + WRITE_SIZE; 64 cycles per
    v_mfma_f64_16x16x4)."""
    import json
    import bench
    root = os.path.join(os.path.dirname(__file__), "..")
    meta = json.load(open(os.path.join(root, "profiles", "r04_pmc_synth400_packed.json")))["_meta"]
    traffic, src, m = bench.pmc_traffic("synth-400", 1, "packed", meta["stored_bytes"])
    assert src == "profiles/r04_pmc_synth400_packed.json" and 1.0 < traffic / meta["stored_bytes"] < 1.5
    ex = bench.pmc_eri_valu("synth-400", 0.027)
    assert ex["profile"] == "profiles/r04_pmc_eri_synth400.json" and 0.2 < ex["valu_issue_utilisation"] < 0.8
    mf = bench.pmc_mfma("synth-400", 8, 8.6e-3)
    assert mf["cycles_per_instruction"] == 64.0 and 0.1 < mf["matrix_pipe_busy_fraction"] < 0.4 and mf["useful_column_fraction"] == 0.5
    assert bench.pmc_mfma("synth-400", 4, 5.0e-3)["kernel"] == "jk_tile_kernel<4, 1, 2>"
