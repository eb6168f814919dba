"""GPU: finite-field electric properties (tuna_amd/properties.py: the reference's dipole / polarisability / hyperpolarisability
drivers, tuna_energy.py:315-650) with the cycles of a property advanced in lockstep -- their densities go through the resident tensor
as one batch per iteration -- against the energies the reference's own cycle gives field by field (tests/golden/field_systems.json,
tools/make_golden.py --field-only) and against the same cycles run one after the other."""
import json
import os
import types

import numpy as np
import pytest

from conftest import GOLD
from oracle import scf_oracle as so
from tuna_amd import molecule as mol
from tuna_amd import properties as props
from tuna_amd.engine import SCF_CONVERGENCE
from tuna_amd.scf import DeviceERI, Integrals

pytestmark = pytest.mark.gpu
FIELDS = json.load(open(os.path.join(GOLD, "field_systems.json")))


def _setup(engine, g):
    atoms = mol.make_atoms(g["symbols"], mol.angstrom_to_bohr(g["R_angstrom"]))
    shells = mol.build_shells(atoms, g["basis"])
    aos = mol.expand_cartesian_aos(shells)
    engine.set_basis(aos).build_eri(True)
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    S, T, V, D, Q = engine.one_electron(xyz, chg, [0.0, 0.0, g["dipole_origin_z"]], spherical=True)
    X, _, _ = engine.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, g["n_occ"])
    ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
    molecule = types.SimpleNamespace(atoms=atoms, n_doubly_occ=g["n_occ"], partition_ranges=ranges)
    calc = types.SimpleNamespace(reference="RHF", DFT_calculation=False, SCF_conv=SCF_CONVERGENCE["extreme"], max_iter=100, DIIS=True,
                                 max_DIIS_matrices=6, damping=True, damping_factor=None, max_damping=0.7, HFX_prop=1.0)
    integrals = Integrals(S, T, V, D, Q, DeviceERI(engine))
    return molecule, calc, integrals, mol.nuclear_repulsion(atoms), X, (P0, P0 / 2, P0 / 2, E0)


@pytest.mark.parametrize("tag", list(FIELDS))
def test_field_energies_and_properties_against_the_reference(engine, tag):
    g = FIELDS[tag]
    molecule, calc, integrals, V_NN, X, guess = _setup(engine, g)
    fe = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched=True, dipole_origin=g["dipole_origin_z"])
    E0 = fe.energies([np.zeros(3)])[0]
    assert abs(E0 - g["energy"]) < 1e-9
    builds0 = integrals.ERI_AO.n_builds
    pol = props.calculate_polarisability(fe, E0)
    builds_pol = integrals.ERI_AO.n_builds - builds0
    for k, e in g["polarisability_energies"].items():
        assert abs(pol["energies"][k] - e) < 1e-9, k
    # second derivatives of energies that agree to ~1e-10 agree to ~1e-10 * 64 / (12 h^2) ~ 5e-4
    assert abs(pol["parallel"] - g["polarisability_parallel"]) < 2e-3
    assert abs(pol["perpendicular"] - g["polarisability_perpendicular"]) < 2e-3
    # eight cycles, ONE batched Fock build per iteration: as many calls as the longest cycle has iterations, not their sum
    assert builds_pol <= 40 and fe.iterations > 4 * builds_pol
    hyp = props.calculate_hyperpolarisability(fe)
    for k, e in g["hyperpolarisability_energies"].items():
        assert abs(hyp["energies"][k] - e) < 1e-9, k
    scale = 1810e-10 / (240 * g["steps"][2] ** 3)                      # what 1e-10 of energy noise does to the eight-point third derivative
    assert abs(hyp["parallel"] - g["hyperpolarisability_parallel"]) < 3 * scale
    assert abs(hyp["perpendicular"] - g["hyperpolarisability_perpendicular"]) < 3 * scale
    # the dipole moment: total = electronic (first derivative) + nuclear about the same origin
    z0 = g["dipole_origin_z"]
    nuclear = sum(float(a.charge) * (a.origin[2] - z0) for a in molecule.atoms)
    h = g["steps"][0]
    Ef, Eb = fe.energies([[0, 0, h], [0, 0, -h]])
    assert abs(Ef - g["dipole_energies"]["+z"]) < 1e-9 and abs(Eb - g["dipole_energies"]["-z"]) < 1e-9
    assert abs(-(Ef - Eb) / (2 * h) - g["electronic_dipole"]) < 2e-5
    # (the polarisability driver's own dipole: central difference of its +z / -z energies at the second-derivative step)
    assert abs(pol["dipole_moment"] - (g["electronic_dipole"] + nuclear)) < 2e-3


def test_lockstep_cycles_equal_cycles_run_one_by_one(engine):
    """The batched driver changes how the work is scheduled, not what a cycle computes: the same energies as eight native cycles
    (tf_scf_rhf with the field term as Fext) run one after the other, with far fewer passes over the tensor."""
    g = FIELDS["co_ccpvdz"]
    molecule, calc, integrals, V_NN, X, guess = _setup(engine, g)
    h = g["steps"][1]
    fields = [[0, 0, 2 * h], [0, 0, h], [0, 0, -h], [0, 0, -2 * h], [2 * h, 0, 0], [h, 0, 0], [-h, 0, 0], [-2 * h, 0, 0]]
    together = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched=True)
    one_by_one = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched=False)
    Ea, Eb = together.energies(fields), one_by_one.energies(fields)
    assert np.abs(np.array(Ea) - np.array(Eb)).max() < 1e-9
    # same trajectories, cycle by cycle.  The counts are not identical: a lockstep iteration builds its Fock matrices two densities per
    # pass (jk_packed_kernel<2>: another order of the same sums than the one-density pass), and a cycle whose |dE| sits on the EXTREME
    # threshold of 1e-11 then stops an iteration earlier or later -- measured: one cycle of the eight, by one (tools/gpu_iteration_counts.py)
    assert abs(together.iterations - one_by_one.iterations) <= 2
    # a field along x mixes AOs of different x parity: the density leaves the block structure of the zero-field problem
    assert abs(Ea[4] - Ea[7]) < 1e-9 and abs(Ea[5] - Ea[6]) < 1e-9                  # E(+x) = E(-x) by symmetry
    # the same batch inside the library (tf_scf_rhf_batch: the native cycle per field on its own host thread and workspace, the Fock
    # builds of an iteration together -- two densities per pass over the tensor)
    native = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched="native")
    b0 = integrals.ERI_AO.n_builds
    Ec = native.energies(fields)
    assert np.abs(np.array(Ec) - np.array(Eb)).max() < 1e-9
    assert abs(native.iterations - one_by_one.iterations) <= 2                      # the same native cycle, only scheduled together
    assert integrals.ERI_AO.n_builds - b0 <= (one_by_one.iterations + 1) // 2 + len(fields)   # passes over the tensor: two densities each
    # an odd number of cycles (the last pass of an iteration carries one density) and a single one
    Ed = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched="native").energies(fields[:3])
    assert np.abs(np.array(Ed) - np.array(Eb[:3])).max() < 1e-9
    Ee = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched="native").energies(fields[:1])
    assert abs(Ee[0] - Eb[0]) < 1e-9


def test_lockstep_batch_reports_cycles_that_do_not_converge(engine):
    """tf_scf_rhf_batch: cycles that run out of iterations end with the reference's message (scf:1435) and their partial results, and
    nobody is left waiting in a Fock build (the call returns); the same batch with enough iterations converges afterwards."""
    from tuna_amd._lib import TunaError
    from tuna_amd.scf import _opts
    g = FIELDS["hf_631g"]
    molecule, calc, integrals, V_NN, X, guess = _setup(engine, g)
    P, _, _, E = guess
    h = g["steps"][1]
    fields = ([0, 0, h], [0, 0, -h], [h, 0, 0])
    terms = [props.apply_electric_field(integrals.D, f) for f in fields]
    o = _opts(calc)
    o["max_iter"] = 3
    with pytest.raises(TunaError, match="not converged in 3 iterations") as e:
        engine.scf_rhf_batch(integrals.S, integrals.T, integrals.V_NE, [P] * 3, [E] * 3, molecule.n_doubly_occ, V_NN, X=X, Fexts=terms,
                             n_atom_ao=molecule.partition_ranges, **o)
    part = e.value.partial
    assert [r["rc"] for r in part] == [-4, -4, -4] and all(r["n_iter"] == 3 and not r["converged"] for r in part)
    o["max_iter"] = 100
    res = engine.scf_rhf_batch(integrals.S, integrals.T, integrals.V_NE, [P] * 3, [E] * 3, molecule.n_doubly_occ, V_NN, X=X, Fexts=terms,
                               n_atom_ao=molecule.partition_ranges, **o)
    ref = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched=False).energies(fields)
    assert all(r["converged"] for r in res) and max(abs(r["energy"] - x) for r, x in zip(res, ref)) < 1e-9


def test_polarisability_input_line(engine):
    """`SPE : F H 0.917 : HF 6-31G : POLAR COREGUESS` -- the reference's keyword (calc:139) and its EXTREME default for second derivatives."""
    from tuna_amd.energy import run
    g = FIELDS["hf_631g"]
    lines = []
    out = run("SPE : F H 0.917 : HF 6-31G : POLAR DIPOLE COREGUESS", silent=False, engine=engine, log=lines.append)
    p = out.properties["polarisability"]
    assert abs(out.energy - g["energy"]) < 1e-8
    assert abs(p["parallel"] - g["polarisability_parallel"]) < 2e-3 and abs(p["perpendicular"] - g["polarisability_perpendicular"]) < 2e-3
    assert abs(p["isotropic"] - (2 * g["polarisability_perpendicular"] + g["polarisability_parallel"]) / 3) < 2e-3
    z0 = g["dipole_origin_z"]
    atoms = mol.make_atoms(g["symbols"], mol.angstrom_to_bohr(g["R_angstrom"]))
    total_ref = g["electronic_dipole"] + sum(float(a.charge) * (a.origin[2] - z0) for a in atoms)   # origin-independent (neutral molecule)
    assert abs(out.properties["dipole_moment"] - total_ref) < 5e-5
    text = "\n".join(lines)
    assert "Isotropic polarisability:" in text and "Total dipole moment:" in text
