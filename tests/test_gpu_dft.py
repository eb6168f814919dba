"""GPU: restricted Kohn-Sham (BASELINE config 4: CO B3LYP/def2-TZVP, plus one case for every other functional of the table: LDA, BLYP,
B3LYP/G, HFS, SVWN3, HFB, BVWN, BVWN3, BHLYP, B1LYP, SLYP) -- AOs on the grid, density,
functional derivatives, V_XC and the whole KS-SCF on the device, against the reference's own tuna_dft.py / tuna_xc.py / tuna_scf.py
run (tests/golden/dft_systems.npz)."""
import numpy as np
import pytest

from conftest import DFT_SYSTEMS
from tuna_amd import molecule as mol

pytestmark = pytest.mark.gpu


def _setup(engine, tag):
    from tuna_amd import dft
    sym, R, basis, nocc, method, grid = DFT_SYSTEMS[tag]
    atoms = mol.make_atoms(sym, R)
    shells = mol.build_shells(atoms, basis)
    aos = mol.expand_cartesian_aos(shells)
    engine.set_basis(aos).build_eri(True)
    pts, wts, info = dft.integration_grid(atoms, grid)
    f = engine.dft_setup(pts, wts, method)
    return atoms, shells, nocc, f


@pytest.mark.parametrize("tag", list(DFT_SYSTEMS))
def test_vxc_of_guess_density(engine, dft_golden, tag):
    g = dft_golden[tag]
    _setup(engine, tag)
    V, n_el, ex, ec = engine.dft_vxc(g["P0"])
    assert abs(n_el - float(g["n_el0"])) < 1e-9
    assert abs(ex - float(g["EX0"])) < 1e-9 and abs(ec - float(g["EC0"])) < 1e-9
    assert np.abs(V - g["V_XC0"]).max() < 1e-9
    assert np.abs(V - V.T).max() == 0.0
    engine.dft_clear()


@pytest.mark.parametrize("tag", list(DFT_SYSTEMS))
def test_kohn_sham_scf_matches_reference(engine, dft_golden, tag):
    from oracle import scf_oracle as so
    g = dft_golden[tag]
    atoms, shells, nocc, f = _setup(engine, tag)
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    S, T, V, _, _ = engine.one_electron(xyz, chg, [0, 0, 0.0])
    X, _, _ = engine.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, nocc)
    ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
    r = engine.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="dynamic", hfx=f["hfx"], n_atom_ao=ranges)
    engine.dft_clear()
    ref = g["table"]
    assert abs(r["energy"] - float(g["energy"])) < 1e-8                       # north-star bar: 1e-8 Eh
    assert abs(r["n_iter"] - len(ref)) <= 1
    n = min(r["n_iter"], len(ref))
    if tag in ("hf_hfb_631g", "hf_bvwn_631g"):
        # HF / 6-31G from the core guess with a pure functional: the first Fock matrix has an exactly degenerate pi pair AT the Fermi level
        # (orbitals 5 and 6: gap 6e-16, tools/gpu_dft_trace.py), so the five occupied orbitals take an arbitrary member of the pair -- the
        # eigensolver's choice.  The quadrature grid is not invariant under rotations about the axis, so that choice moves the next
        # energies by ~1e-7..1e-5 Eh (in the reference as well) until the occupation is closed-shell again: only the damping factors,
        # the first energy and the converged state compare.
        assert abs(r["table"][0, 1] - ref[0, 1]) < 5e-8
        np.testing.assert_allclose(r["table"][:n, 1], ref[:n, 1], atol=1e-4)
    else:
        np.testing.assert_allclose(r["table"][:n, 1], ref[:n, 1], atol=5e-8)
    np.testing.assert_allclose(r["table"][:n, 6], ref[:n, 6], atol=1e-6)
    np.testing.assert_allclose(r["components"][:5], g["components"], atol=1e-7)
    np.testing.assert_allclose(r["epsilons"], g["eps"], atol=1e-6)


def test_b3lyp_input_line(dft_golden):
    """`SPE : C O 1.128 : B3LYP DEF2-TZVP` -- config 4 through the input-line entry (SAD guess, default grid and thresholds):
    converged energy within the medium thresholds of the reference's EXTREME-converged value."""
    from tuna_amd.energy import run
    out = run("SPE : C O 1.128 : B3LYP DEF2-TZVP")
    assert abs(out.energy - float(dft_golden["c4_co_b3lyp_def2tzvp"]["energy"])) < 2e-6
    out = run("SPE : C O 1.128 : B3LYP DEF2-TZVP : EXTREME COREGUESS")
    assert abs(out.energy - float(dft_golden["c4_co_b3lyp_def2tzvp"]["energy"])) < 1e-8
