"""Finite-field electric properties on the device-resident tensor: the reference's calculate_numerical_dipole_moment
(tuna_energy.py:565-650), calculate_polarisability (energy:315-425) and calculate_hyperpolarisability (energy:430-560).

The reference evaluates the molecular energy once per field value, one self-consistent field cycle after the other, each
building its Fock matrices from the same dense tensor.  Two ways to run the 2, 8 or 12 cycles of a property here:
  * batched=False (the default): the native cycle (tf_scf_rhf, everything on the device) once per field value on the resident
    tensor -- the fastest path measured (tools/gpu_field_timing.py: eight cycles at N = 400 in 1.17 s);
  * batched=True: the cycles in LOCKSTEP (tuna_amd.scf.run_cycles_in_lockstep): the densities of all cycles still iterating go
    through the tensor as one batch per iteration (pairs of densities per pass).  The Fock builds are then as cheap as they can be,
    but these cycles are orchestrated from the host (NumPy algebra, one eigensolve per iteration), which costs more than the
    batching saves (4.0 s for the same eight cycles): the mechanism the batching needs, kept for the native lockstep cycle that
    would make it pay.
Every cycle follows the reference's own iteration order and stops by its own convergence test, so the energies -- and the
finite-difference formulas restated from tuna_util.py:581-680 -- give the reference's numbers.  Restricted Hartree-Fock only.
"""
from __future__ import annotations

import numpy as np

from . import scf as scf_mod
from ._lib import TunaError

FIRST_ELEC_DERIVATIVE_STEP = 0.00001     # tuna_util.py:77
SECOND_ELEC_DERIVATIVE_STEP = 0.001      # tuna_util.py:85
THIRD_ELEC_DERIVATIVE_STEP = 0.0015      # tuna_util.py:93


def calculate_first_derivative(F_m_1, F_p_1, dx):                                   # tuna_util.py:581-601
    return (F_p_1 - F_m_1) / (2 * dx)


def calculate_second_derivative(F_m_2, F_m_1, F, F_p_1, F_p_2, dx):                 # tuna_util.py:612-638
    return (-F_m_2 + 16 * F_m_1 - 30 * F + 16 * F_p_1 - F_p_2) / (12 * dx ** 2)


def calculate_third_derivative(F_m_4, F_m_3, F_m_2, F_m_1, F_p_1, F_p_2, F_p_3, F_p_4, dx):   # tuna_util.py:648-673
    return (-7 * F_m_4 + 72 * F_m_3 - 338 * F_m_2 + 488 * F_m_1 - 488 * F_p_1 + 338 * F_p_2 - 72 * F_p_3 + 7 * F_p_4) / (240 * dx ** 3)


def apply_electric_field(D, electric_field):                                         # tuna_kernel.py:660-677
    return np.einsum("i,ijk->jk", np.asarray(electric_field, dtype=float), D, optimize=True)


def calculate_nuclear_dipole_moment(dipole_origin, charges, coordinates):            # tuna_props.py:28-51
    return sum((coordinates[i][2] - dipole_origin) * charges[i] for i in range(len(charges)))


class FieldEnergies:
    """Energies of a list of electric fields for one molecule: `energies(fields)` runs the cycles in lockstep inside the library
    (batched="native": tf_scf_rhf_batch), in lockstep orchestrated from the host (batched=True), or one after the other through the
    native cycle (batched=False, the reference's order of work); "auto" picks the native lockstep for N >= 96.  Counts the tensor
    passes either way."""

    def __init__(self, molecule, calculation, integrals, V_NN, X, guess_objects, batched="auto", dipole_origin=None):
        if getattr(calculation, "reference", "RHF") == "UHF" or getattr(calculation, "DFT_calculation", False):
            raise TunaError("finite-field properties are available for restricted Hartree-Fock in this build")
        self.molecule, self.calculation, self.integrals = molecule, calculation, integrals
        self.V_NN, self.X, self.guess, self.batched = V_NN, X, guess_objects, batched
        self.base_field = np.array(getattr(calculation, "electric_field", np.zeros(3)), dtype=float)
        self.dipole_origin = dipole_origin       # z of the origin the dipole integrals were taken about (None: the centre of mass, kernel:312)
        self.cycles = 0
        self.iterations = 0

    def energies(self, fields):
        terms = [apply_electric_field(self.integrals.D, self.base_field + np.asarray(f, dtype=float)) for f in fields]
        eng = getattr(self.integrals.ERI_AO, "engine", None)
        mode = self.batched
        if mode == "auto":
            # the lockstep batch pays where the O(N^3) steps of one cycle leave most of the device idle and a pass over the tensor costs
            # something (N = 400: 0.58 s against 1.15 s for the eight cycles of a polarisability); tiny problems are launch-bound and
            # run faster one by one (CO/cc-pVDZ: 45 ms against 68 ms)
            mode = "native" if (eng is not None and eng.N >= 96) else False
        if mode == "native" and (eng is None or eng.world > 1):
            mode = False                                     # (sharded tensors: the cycles run one by one, each with its all-reduce)
        if mode:
            lock = scf_mod.run_cycles_in_native_lockstep if mode == "native" else scf_mod.run_cycles_in_lockstep
            res = lock(self.molecule, self.calculation, self.integrals, self.V_NN, self.X, self.guess, terms)
            out = [r["energy"] for r in res]
            self.iterations += sum(r["n_iter"] for r in res)
        else:
            out = []
            F_keep = self.integrals.F
            try:
                for t in terms:
                    self.integrals.F = t
                    o = scf_mod.run_self_consistent_field_cycle(self.molecule, self.calculation, self.integrals, self.V_NN, self.X,
                                                                self.guess, None, True)
                    out.append(o.energy)
                    self.iterations += o.n_iterations
            finally:
                self.integrals.F = F_keep
        self.cycles += len(terms)
        return out


def _geometry(fe):
    from . import guess as guess_mod
    atoms = fe.molecule.atoms
    origin = fe.dipole_origin if fe.dipole_origin is not None else (guess_mod.centre_of_mass(atoms) if len(atoms) == 2 else 0.0)
    return origin, [float(a.charge) for a in atoms], [a.origin for a in atoms]


def calculate_numerical_dipole_moment(field_energies: FieldEnergies, silent=True, log=print):
    """energy:565-650: -dE/dF_z by central differences + the nuclear dipole about the centre of mass."""
    h = FIRST_ELEC_DERIVATIVE_STEP
    z = np.array([0.0, 0.0, h])
    if not silent:
        log("\n Beginning dipole moment calculation... ")
        log(f"  Using a finite field magnitude of {h:.5f} au.")
    E_forward, E_backward = field_energies.energies([z, -z])
    electronic = -1 * calculate_first_derivative(E_backward, E_forward, h)
    nuclear = calculate_nuclear_dipole_moment(*_geometry(field_energies))
    total = electronic + nuclear
    if not silent:
        log(f"\n  Nuclear dipole moment:                 {nuclear:10.5f}")
        log(f"  Electronic dipole moment:              {electronic:10.5f}")
        log(f"\n  Total dipole moment:                   {total:10.5f}")
    return total


def calculate_polarisability(field_energies: FieldEnergies, energy, silent=True, log=print):
    """energy:315-425: the parallel and perpendicular second derivatives (five-point stencils: eight field evaluations, ONE batch)
    -> dict(parallel, perpendicular, anisotropic, isotropic, dipole_moment, energies)."""
    h = SECOND_ELEC_DERIVATIVE_STEP
    x, z = np.array([h, 0.0, 0.0]), np.array([0.0, 0.0, h])
    if not silent:
        log("\n Beginning dipole-dipole polarisability calculation... ")
        log(f"  Using a finite field magnitude of {h:.5f} au.")
    fields = [2 * z, z, -z, -2 * z, 2 * x, x, -x, -2 * x]       # the order of energy:359-373, parallel then perpendicular
    E = field_energies.energies(fields)
    parallel = -1 * calculate_second_derivative(E[3], E[2], energy, E[1], E[0], h)
    perpendicular = -1 * calculate_second_derivative(E[7], E[6], energy, E[5], E[4], h)
    electronic = -1 * calculate_first_derivative(E[2], E[1], h)
    total_dipole = electronic + calculate_nuclear_dipole_moment(*_geometry(field_energies))
    anisotropic = parallel - perpendicular
    isotropic = (perpendicular * 2 + parallel) / 3
    if not silent:
        log(f"\n  Dipole moment:                         {total_dipole:10.4f}")
        log(f"\n  Ansotropic polarisability:             {anisotropic:10.4f}")          # (sic, energy:419)
        log(f"  Isotropic polarisability:              {isotropic:10.4f}")
    return dict(parallel=parallel, perpendicular=perpendicular, anisotropic=anisotropic, isotropic=isotropic, dipole_moment=total_dipole,
                energies=dict(zip(["+2z", "+z", "-z", "-2z", "+2x", "+x", "-x", "-2x"], E)))


def calculate_hyperpolarisability(field_energies: FieldEnergies, silent=True, log=print):
    """energy:430-560: parallel (eight-point third derivative along z) and perpendicular (mixed x, x, z) components: twelve field
    evaluations, ONE batch -> dict(parallel, perpendicular, dipole_moment, energies)."""
    h = THIRD_ELEC_DERIVATIVE_STEP
    x, z = np.array([h, 0.0, 0.0]), np.array([0.0, 0.0, h])
    if not silent:
        log("\n Beginning hyperpolarisability calculation... ")
        log(f"  Using a finite field magnitude of {h:.5f} au.")
    names = ["+3z", "+2z", "+z", "-z", "-2z", "-3z", "-4z", "+4z", "+x+z", "-x+z", "+x-z", "-x-z"]
    fields = [3 * z, 2 * z, z, -z, -2 * z, -3 * z, -4 * z, 4 * z, x + z, -x + z, x - z, -x - z]
    E = dict(zip(names, field_energies.energies(fields)))
    parallel = -1 * calculate_third_derivative(E["-4z"], E["-3z"], E["-2z"], E["-z"], E["+z"], E["+2z"], E["+3z"], E["+4z"], h)
    perpendicular = -(E["-x+z"] - 2 * E["+z"] + E["+x+z"] - E["-x-z"] + 2 * E["-z"] - E["+x-z"]) / (2 * h ** 3)
    electronic = -1 * calculate_first_derivative(E["-z"], E["+z"], h)
    total_dipole = electronic + calculate_nuclear_dipole_moment(*_geometry(field_energies))
    if not silent:
        log(f"\n  Dipole moment:                         {total_dipole:10.4f}")
        log(f"\n  Parallel hyperpolarisability:          {parallel:10.4f}")
        log(f"  Perpendicular hyperpolarisability:     {perpendicular:10.4f}")
    return dict(parallel=parallel, perpendicular=perpendicular, dipole_moment=total_dipole, energies=E)
