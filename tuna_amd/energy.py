"""Energy driver: the hot-path subset of tuna_energy.calculate_energy (energy:875-964) and tuna.py's input line.

    from tuna_amd.energy import run
    out = run("SPE : N N 1.0977 : HF CC-PVTZ : EXTREME NODAMP")

Input line format `TYPE : A B R : METHOD BASIS : KEYWORDS` (tuna.py:87-99).  Supported here: TYPE = SPE, METHOD = HF
(restricted) or UHF / any multiplicity via ML (unrestricted), the basis sets shipped in tuna_amd/data, and the SCF keywords of SURVEY.md section 5
(LOOSE/MEDIUM/TIGHT/EXTREME, MAXITER n, DIIS [n]/NODIIS, DAMP x/NODAMP/MAXDAMP x, SLOWCONV/VERYSLOWCONV, HFX x,
CARTHARM, DECONTRACT, COREGUESS/SADGUESS, CH n, ML n).  Everything numerical runs on the GPU through the C ABI.  The initial
guess is the reference's default for single points, the superposition of atomic densities (tuna_amd/guess.py).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from . import molecule as mol
from ._lib import TunaError
from .engine import SCF_CONVERGENCE, Engine
from .scf import DeviceERI, Integrals, construct_density_matrix, run_self_consistent_field_cycle


@dataclass
class Calculation:
    """The fields of the reference's Calculation (tuna_calc.py:532-596) that the hot path reads."""
    calculation_type: str = "SPE"
    method: str = "HF"
    basis: str = "STO-3G"
    reference: str = "RHF"
    charge: int = 0
    SCF_conv: dict = field(default_factory=lambda: SCF_CONVERGENCE["medium"])
    max_iter: int = 100                 # MAXITER, calc:158
    DIIS: bool = True                   # calc:198
    max_DIIS_matrices: int = 6
    damping: bool = True                # calc:199
    damping_factor: float | None = None
    max_damping: float = 0.7            # calc:159
    HFX_prop: float = 1.0               # calc:207
    cartesian_harmonics: bool = False   # CARTHARM, calc:91
    decontract: bool = False            # DECONTRACT, calc:90
    core_guess: bool = False            # COREGUESS, calc:95; default is the superposition of atomic densities
    DFT_calculation: bool = False
    multiplicity: int = 1               # ML, calc:151
    functional: str | None = None       # Kohn-Sham functional name (tuna_amd.dft.FUNCTIONALS) or None for Hartree-Fock
    grid_conv: str = "medium"           # LOOSEGRID/MEDIUMGRID/TIGHTGRID..., util:129-137
    X_alpha: float = 2 / 3              # XA, calc:156
    dipole: bool = False                # DIPOLE, calc:137: finite-field electric properties (tuna_amd/properties.py)
    polarisability: bool = False        # POLAR, calc:139
    hyperpolarisability: bool = False   # HYPER, calc:140
    electric_field: tuple = (0.0, 0.0, 0.0)            # EX, EY, EZ, calc:160-162, 431
    electric_field_gradient: tuple = (0.0, 0.0, 0.0)   # EGX, EGY, EGZ, calc:163-165, 432
    S_eigenvalue_threshold: float = 1e-7               # STHRESH, calc:157
    number_of_threads: int = 4                         # THREADS, calc:153 (host threads of the reference's OpenMP loops: no meaning here)


@dataclass
class Molecule:
    atoms: list
    shells: list
    aos: mol.AOList
    n_electrons: int
    n_doubly_occ: int
    n_alpha: int
    n_beta: int
    partition_ranges: list
    n_basis: int
    n_cartesian_basis: int


def parse_input(input_line: str):
    """tuna.py:59-161 -> (calculation_type, method, basis, atomic_symbols, R_bohr | None, params)."""
    try:
        sections = input_line.upper().strip().split(":")
        calculation_type = sections[0].strip()
        geometry = [g for g in sections[1].strip().split(" ") if g.strip()]
        method, basis = sections[2].strip().split()
        params = sections[3].strip().split() if len(sections) == 4 else []
    except Exception:
        raise TunaError("Input line formatted incorrectly! Read the manual for help.")
    symbols = geometry[:2] if len(geometry) >= 3 else geometry[:1]
    try:
        R = [float(x) for x in geometry[len(symbols):]]
    except ValueError:
        raise TunaError("Could not parse bond length!")
    if len(symbols) == 2 and len(R) != 1:
        raise TunaError("Two atoms requested without a bond length!")
    if R and R[0] < 0.01:
        raise TunaError(f"Bond length ({R[0]} angstroms) is too small! Minimum bond length is 0.01 angstroms.")
    return calculation_type, method, basis, symbols, (mol.angstrom_to_bohr(R[0]) if R else None), params


def interpret_keywords(params, calc: Calculation) -> Calculation:
    """The SCF subset of tuna_calc.py:83-217, 357-521."""
    it = iter(range(len(params)))
    custom = {}
    for k in it:
        p = params[k]

        def value():
            try:
                next(it)
                return params[k + 1]
            except (StopIteration, IndexError):
                raise TunaError(f"Keyword {p} needs a value")
        if p in ("LOOSE", "MEDIUM", "TIGHT", "EXTREME"):
            calc.SCF_conv = SCF_CONVERGENCE[p.lower()]
        elif p == "MAXITER":
            calc.max_iter = int(value())
        elif p == "NODIIS":
            calc.DIIS = False
        elif p == "DIIS":
            if k + 1 < len(params) and params[k + 1].isdigit():
                calc.max_DIIS_matrices = int(value())
        elif p == "NODAMP":
            calc.damping = False
        elif p == "DAMP":
            calc.damping, calc.damping_factor = True, float(value())
        elif p == "SLOWCONV":
            calc.damping, calc.damping_factor = True, 0.5
        elif p == "VERYSLOWCONV":
            calc.damping, calc.damping_factor = True, 0.85
        elif p == "MAXDAMP":
            calc.max_damping = float(value())
        elif p == "HFX":
            calc.HFX_prop = float(value())
        elif p == "CARTHARM":
            calc.cartesian_harmonics = True
        elif p == "DECONTRACT":
            calc.decontract = True
        elif p in ("CH", "CHARGE"):
            calc.charge = int(value())
        elif p in ("ML", "MULTIPLICITY"):
            calc.multiplicity = int(value())
        elif p in ("LOOSEGRID", "MEDIUMGRID", "TIGHTGRID", "EXTREMEGRID"):
            calc.grid_conv = p[:-4].lower()
        elif p == "XA":
            calc.X_alpha = float(value())
        elif p == "COREGUESS":
            calc.core_guess = True
        elif p == "DIPOLE":
            calc.dipole = True
        elif p in ("POLAR", "POLARISABILITY", "POLARIZABILITY"):
            calc.polarisability = True
        elif p in ("HYPER", "HYPERPOLARISABILITY", "HYPERPOLARIZABILITY"):
            calc.hyperpolarisability = True
        elif p == "THREADS":
            calc.number_of_threads = int(value())           # accepted and ignored: the integrals run on the GPU
        elif p == "STHRESH":
            calc.S_eigenvalue_threshold = float(value())
        elif p in ("EX", "EY", "EZ"):
            f = list(calc.electric_field)
            f["XYZ".index(p[1])] = float(value())
            calc.electric_field = tuple(f)
        elif p in ("EGX", "EGY", "EGZ"):
            f = list(calc.electric_field_gradient)
            f["XYZ".index(p[2])] = float(value())
            calc.electric_field_gradient = tuple(f)
        elif p in ("ECONV", "RMSDP", "MAXDP", "DIISERR"):      # calc:187-190, applied over the named criteria at calc:491-494
            custom[{"ECONV": "delta_E", "RMSDP": "RMS_DP", "MAXDP": "max_DP", "DIISERR": "commutator"}[p]] = float(value())
        elif p in ("SADGUESS", "SCFGUESS", "T", "P", "DEBUG"):
            pass                                             # (SCFGUESS: the reference's default guess path, calc:405-421 -- reproduced by default)
        else:
            raise TunaError(f"Keyword \"{p}\" is not supported on the GPU hot path (SCF keywords only)")
    # calc:473-494: first the named criteria -- a derivative request without LOOSE..EXTREME tightens them (polarisabilities: extreme,
    # dipole: tight) -- then ECONV / RMSDP / MAXDP / DIISERR are applied over whichever set was chosen
    if not any(p in params for p in ("LOOSE", "MEDIUM", "TIGHT", "EXTREME")):
        if calc.polarisability or calc.hyperpolarisability:
            calc.SCF_conv = SCF_CONVERGENCE["extreme"]
        elif calc.dipole:
            calc.SCF_conv = SCF_CONVERGENCE["tight"]
    if custom:
        calc.SCF_conv = dict(calc.SCF_conv, **custom)
    return calc


def build_molecule_and_integrals(symbols, R_bohr, calc: Calculation, engine: Engine, sharded_fock_factory=None):
    """energy:770-870 for the hot path: molecule, one- and two-electron integrals (GPU), orthogonaliser, guess."""
    atoms = mol.make_atoms(symbols, R_bohr)
    shells = mol.build_shells(atoms, calc.basis, calc.decontract)
    aos = mol.expand_cartesian_aos(shells)
    spherical = not calc.cartesian_harmonics
    n_el = mol.electron_count(atoms, calc.charge)
    if n_el <= 0:
        raise TunaError("Zero electrons specified!" if n_el == 0 else "Negative number of electrons specified!")
    n_unpaired = calc.multiplicity - 1
    if calc.multiplicity < 1 or (n_el - n_unpaired) % 2 or n_unpaired > n_el:
        raise TunaError("Impossible charge and multiplicity combination!")
    n_alpha, n_beta = (n_el + n_unpaired) // 2, (n_el - n_unpaired) // 2
    if calc.multiplicity != 1:
        calc.reference = "UHF"                               # tuna_molecule.py:307
    timings = {}
    t0 = time.perf_counter()
    engine.set_basis(aos)
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    from . import guess as guess_mod
    com = [0.0, 0.0, guess_mod.centre_of_mass(atoms) if len(atoms) == 2 else 0.0]     # dipole origin, kernel:312
    S, T, V, D, Q = engine.one_electron(xyz, chg, com, spherical=spherical)
    timings["One-electron integrals"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    engine.build_eri(spherical=spherical)
    timings["Two-electron integrals"] = time.perf_counter() - t0
    fock = sharded_fock_factory(engine) if sharded_fock_factory is not None and engine.world > 1 else None
    integrals = Integrals(S, T, V, D, Q, DeviceERI(engine, fock))
    if calc.functional is not None:                         # Kohn-Sham: grid + AOs on the grid (tuna_dft.py:94-208)
        from . import dft as dft_mod
        if calc.reference == "UHF":
            raise TunaError("unrestricted Kohn-Sham is not on the GPU path in this build")
        t0 = time.perf_counter()
        pts, wts, ginfo = dft_mod.integration_grid(atoms, calc.grid_conv)
        f = engine.dft_setup(pts, wts, calc.functional, calc.X_alpha)
        calc.HFX_prop, calc.DFT_calculation = f["hfx"], True
        timings["Integration grid setup"] = time.perf_counter() - t0
    else:
        engine.dft_clear()
    dim = (lambda s: s.n_sph) if spherical else (lambda s: s.n_cart)
    ranges = [sum(dim(s) for s in shells if s.atom == a) for a in range(len(atoms))]
    molecule = Molecule(atoms, shells, aos, n_el, n_el // 2, n_alpha, n_beta, ranges, engine.N, aos.n)
    t0 = time.perf_counter()
    X, smallest, S_inv = engine.orthogonaliser(S)
    timings["Fock orthogonalisation matrix"] = time.perf_counter() - t0
    if smallest < calc.S_eigenvalue_threshold:            # STHRESH, kernel:887
        raise TunaError("An overlap matrix eigenvalue is too small! Change the basis set or decrease the threshold with STHRESH.")
    t0 = time.perf_counter()
    use_core = calc.core_guess or calc.cartesian_harmonics or any(a.charge == 0 for a in atoms)
    if use_core:
        _, C0 = engine.diagonalise(integrals.H_core, X)
        if calc.reference == "UHF":
            Pa0, Pb0 = construct_density_matrix(C0, n_alpha, 1), construct_density_matrix(C0, n_beta, 1)
            P0 = Pa0 + Pb0
        else:
            P0 = construct_density_matrix(C0, molecule.n_doubly_occ, 2)
            Pa0 = Pb0 = P0 / 2
        E0 = float(np.einsum("mn,mn->", integrals.H_core, P0))     # guess energy, tuna_guess.py:429
    else:
        P0, Pa0, Pb0, E0 = guess_mod.superposition_guess(engine, atoms, S, S_inv, engine.sph_matrix(), n_alpha, n_beta, integrals.H_core)
    timings["Initial guess"] = time.perf_counter() - t0
    return molecule, integrals, X, (P0, Pa0, Pb0, E0), timings


def calculate_energy(symbols, R_bohr, calc: Calculation, engine: Engine | None = None, silent=True, log=print):
    own = engine is None
    engine = engine or Engine(0)
    try:
        molecule, integrals, X, guess, timings = build_molecule_and_integrals(symbols, R_bohr, calc, engine)
        V_NN = mol.nuclear_repulsion(molecule.atoms)
        # field terms of the Hamiltonian, set after the guess as the reference does (energy:915-919; kernel:660-707: only two
        # independent components of the quadrupole tensor are used, [Q0, Q0, Q1])
        if np.linalg.norm(calc.electric_field) > 0:
            integrals.F = np.einsum("i,ijk->jk", np.asarray(calc.electric_field, dtype=float), integrals.D, optimize=True)
        if np.linalg.norm(calc.electric_field_gradient) > 0:
            Q3 = np.array([integrals.Q[0], integrals.Q[0], integrals.Q[1]])
            integrals.G = np.einsum("i,ijk->jk", np.asarray(calc.electric_field_gradient, dtype=float), Q3, optimize=True)
        if not silent:
            log(f" Nuclear repulsion energy: {V_NN:.10f}\n")
        t0 = time.perf_counter()
        out = run_self_consistent_field_cycle(molecule, calc, integrals, V_NN, X, guess, None, silent, log)
        timings["Self-consistent field"] = time.perf_counter() - t0
        out.timings.update(timings)
        if not silent:
            if calc.functional is not None:
                space = " " * max(0, 8 - len(calc.functional))
                log(f"\n Restricted {calc.functional} energy: {space}      " + f"{out.energy:16.10f}")     # kernel:854
            else:
                label = "\n Unrestricted Hartree-Fock energy: " if calc.reference == "UHF" else "\n Restricted Hartree-Fock energy:   "
                log(label + f"{out.energy:16.10f}")                                      # kernel:846-850
        if calc.method == "MP2":
            # second-order Moller-Plesset correlation energy on the device-resident tensor (tuna_mp.py:834-906; all-electron,
            # SURVEY.md section 8d config 5): AO->MO of the (ia|jb) block + the energy sums, tf_mp2_rhf
            if calc.reference == "UHF":
                raise TunaError("MP2 is available for a restricted reference only in this build.")
            t0 = time.perf_counter()
            r = engine.mp2_rhf(out.molecular_orbitals, out.epsilons, molecule.n_doubly_occ, 0)
            out.timings["MP2 energy"] = time.perf_counter() - t0
            out.mp2 = r
            E_SCF = out.energy
            out.correlation_energy_mp2 = r["E_MP2"]
            out.energy = E_SCF + r["E_MP2"]
            if not silent:
                log(f"\n  Same spin contribution:             {r['E_SS']:13.10f}")       # mp:904-906
                log(f"  Opposite spin contribution:         {r['E_OS']:13.10f}")
                log(f"\n  MP2 correlation energy:             {r['E_MP2']:13.10f}")
        if not silent:
            log(" Final single point energy:        " + f"{out.energy:16.10f}")        # kernel:1305
        if calc.dipole or calc.polarisability or calc.hyperpolarisability:
            # finite-field properties (energy:941-957): the cycles of a property run in lockstep on the resident tensor
            from . import properties as props
            if calc.method == "MP2":
                raise TunaError("finite-field properties are available for Hartree-Fock energies in this build")
            t0 = time.perf_counter()
            fe = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess)
            out.properties = {}
            if calc.dipole:
                out.properties["dipole_moment"] = props.calculate_numerical_dipole_moment(fe, silent, log)
            if calc.polarisability:
                out.properties["polarisability"] = props.calculate_polarisability(fe, out.energy, silent, log)
            if calc.hyperpolarisability:
                out.properties["hyperpolarisability"] = props.calculate_hyperpolarisability(fe, silent, log)
            out.timings["Electric properties"] = time.perf_counter() - t0
        out.integrals = integrals if not own else None      # the device tensor dies with an engine we own
        return out
    finally:
        if own:
            engine.close()


def run(input_line: str, silent: bool = True, engine: Engine | None = None, log=print):
    """tuna.py:345 `run(input_line, suppress_output)` for single-point restricted Hartree-Fock."""
    ctype, method, basis, symbols, R, params = parse_input(input_line)
    if ctype != "SPE":
        raise TunaError(f"Calculation type \"{ctype}\" is not supported.")
    from . import dft as dft_mod
    if method not in ("HF", "RHF", "UHF", "MP2", "RMP2") and method not in dft_mod.FUNCTIONALS:
        raise TunaError(f"Electronic structure method \"{method}\" is not supported.")
    calc = interpret_keywords(params, Calculation(ctype, method if method in dft_mod.FUNCTIONALS else ("MP2" if "MP2" in method else "HF"), basis))
    if method == "UHF":
        calc.reference = "UHF"
    if method in dft_mod.FUNCTIONALS:
        calc.functional = method
    return calculate_energy(symbols, R, calc, engine, silent, log)
