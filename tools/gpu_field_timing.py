"""GPU: wall time of the eight field evaluations of a polarisability (tuna_amd/properties.py), cycles in lockstep with batched Fock
builds (host-orchestrated) against eight native cycles one after the other.  usage: python tools/gpu_field_timing.py [native-only]
(TF_ERI_LAYOUT=t: on the tiles layout, whose lockstep iterations send all eight densities through the tensor in one wide pass)"""
import os, sys, time, types
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from oracle import scf_oracle as so
from tuna_amd import molecule as mol, properties as props
from tuna_amd.engine import Engine, SCF_CONVERGENCE
from tuna_amd.scf import DeviceERI, Integrals

def case(eng, atoms, shells, aos, nocc, label, conv):
    eng.set_basis(aos).build_eri(True)
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    S, T, V, D, Q = eng.one_electron(xyz, chg, [0.0, 0.0, 0.5 * atoms[-1].origin[2]], spherical=True)
    X, _, _ = eng.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, nocc)
    ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
    molecule = types.SimpleNamespace(atoms=atoms, n_doubly_occ=nocc, partition_ranges=ranges)
    calc = types.SimpleNamespace(reference="RHF", DFT_calculation=False, SCF_conv=SCF_CONVERGENCE[conv], max_iter=100, DIIS=True,
                                 max_DIIS_matrices=6, damping=True, damping_factor=None, max_damping=0.7, HFX_prop=1.0)
    integrals = Integrals(S, T, V, D, Q, DeviceERI(eng))
    V_NN = mol.nuclear_repulsion(atoms)
    h = props.SECOND_ELEC_DERIVATIVE_STEP
    fields = [[0, 0, 2 * h], [0, 0, h], [0, 0, -h], [0, 0, -2 * h], [2 * h, 0, 0], [h, 0, 0], [-h, 0, 0], [-2 * h, 0, 0]]
    modes = ("native", "native", False) if "native-only" in sys.argv else ("native", True, False, "native", False)
    for batched in modes:
        fe = props.FieldEnergies(molecule, calc, integrals, V_NN, X, (P0, P0 / 2, P0 / 2, E0), batched=batched)
        b0 = integrals.ERI_AO.n_builds
        t0 = time.perf_counter(); E = fe.energies(fields); dt = time.perf_counter() - t0
        print(f"{label} N={eng.N} {('native lockstep (tf_scf_rhf_batch)' if batched == 'native' else 'host-orchestrated lockstep') if batched else 'native, one by one'}: {dt * 1e3:8.1f} ms, {fe.iterations} iterations, "
              f"{integrals.ERI_AO.n_builds - b0} host-level Fock calls, E(+2z) = {E[0]:.10f}", flush=True)

with Engine(0) as eng:
    atoms = mol.make_atoms(["C", "O"], mol.angstrom_to_bohr(1.128)); shells = mol.build_shells(atoms, "cc-pVDZ")
    case(eng, atoms, shells, mol.expand_cartesian_aos(shells), 7, "CO/cc-pVDZ", "extreme")
    atoms, shells, aos, nocc, desc = bench.build_workload("synth-400")
    case(eng, atoms, shells, aos, nocc, "synth-400", "tight")
