"""Drop-in face of the reference's SCF functions (seam 2): the names, argument order and return
values of TUNA/tuna_scf.py ("scf"), with the work done by libtunafock through the C ABI.

* `integrals.ERI_AO` is a `DeviceERI` handle -- the tensor lives in HBM (rows i>=j, sharded over ranks), not in host RAM.
* `calculate_coulomb_matrix` / `calculate_exchange_matrix` (scf:55-72, 27-44) are one fused device pass (cached per
  density, so the reference's two calls cost one build); with several ranks the partial [J;K] is all-reduced (RCCL).
* `run_self_consistent_field_cycle` (scf:1292-1435) runs the whole RHF cycle natively (`tf_scf_rhf`: HIP J/K + rocBLAS +
  rocSOLVER) on one GPU; with a sharded tensor it runs the same iteration order in Python with device Fock builds.
* The small helpers keep the reference's NumPy form where they are O(N^2) bookkeeping (`symmetrise`, `calculate_SCF_changes`,
  `check_convergence`, energies as traces); `diagonalise_Fock_matrix` goes through rocSOLVER (`tf_orthogonaliser` family).
There is no CPU fallback for the tensor contractions: a NumPy ERI array is rejected.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from ._lib import TunaError
from .engine import SCF_CONVERGENCE, Engine


def symmetrise(matrix):                                   # tuna_util.py:748-764
    return (1 / 2) * (matrix + matrix.T)


class DeviceERI:
    """Handle of the HBM-resident (ij|kl) tensor built by `Engine.build_eri` -- what `integrals.ERI_AO` holds here."""

    def __init__(self, engine: Engine, sharded_fock=None):
        self.engine = engine
        self.shape = (engine.N,) * 4
        self._fock = sharded_fock           # tuna_amd.distributed.ShardedFock when world > 1 (built on first use otherwise)
        self._key = None
        self._jk = None
        self.n_builds = 0

    def jk(self, P):
        P = np.ascontiguousarray(P, dtype=np.float64)
        key = (P.shape, P.tobytes())
        if key != self._key:
            if self.engine.world > 1 and self._fock is None:
                # a rank's own pass gives PARTIAL sums only: never hand those out as J and K
                from . import distributed as tdist
                import torch.distributed as dist
                if not (dist.is_available() and dist.is_initialized()):
                    raise TunaError(f"the tensor is sharded over {self.engine.world} ranks but torch.distributed is not initialised: "
                                    "a Fock build needs the all-reduce of the partial [J;K]")
                self._fock = tdist.ShardedFock(self.engine)
            self._jk = self._fock(P) if self._fock is not None else self.engine.fock_jk(P)
            self._key = key
            self.n_builds += 1
        return self._jk

    def dense(self):
        """The reference's dense float64[N,N,N,N] (for post-SCF consumers); moderate N only, one rank only (a rank of a sharded
        tensor holds zeros for the rows of the others)."""
        if self.engine.world > 1:
            raise TunaError("DeviceERI.dense(): the tensor is sharded over several ranks; sum engine.copy_eri() over the ranks instead")
        return self.engine.copy_eri()


def _device(ERI_AO) -> DeviceERI:
    if not isinstance(ERI_AO, DeviceERI):
        raise TunaError("tuna_amd contracts the ERI tensor on the GPU only: pass the DeviceERI handle (integrals.ERI_AO), "
                        "not a NumPy array -- there is no CPU fallback")
    return ERI_AO


def calculate_exchange_matrix(P, ERI_AO):                 # scf:27-44   K = einsum("ilkj,kl->ij")
    return _device(ERI_AO).jk(P)[1]


def calculate_coulomb_matrix(P, ERI_AO):                  # scf:55-72   J = einsum("ijkl,kl->ij")
    return _device(ERI_AO).jk(P)[0]


def construct_density_matrix(molecular_orbitals, n_occ, n_electrons_per_orbital):   # scf:183-211
    occupied_mos = molecular_orbitals[:, :n_occ]
    return symmetrise(n_electrons_per_orbital * occupied_mos @ occupied_mos.T)


def diagonalise_Fock_matrix(F, X, engine: Engine | None = None):                     # scf:222-250
    """eigh(sym(X^T F X)) through rocSOLVER (native RHF uses the same routine); returns (epsilons, molecular_orbitals)."""
    from .integral import default_engine
    eng = engine or default_engine()
    return eng.diagonalise(F, X)


def calculate_SCF_changes(E, E_old, P, P_old):                                        # scf:261-288
    delta_P = P - P_old
    return E - E_old, np.max(np.abs(delta_P)), np.mean(delta_P ** 2) ** (1 / 2)


def check_convergence(SCF_conv, step, delta_E, max_DP, RMS_DP, commutator, calculation=None, silent=False):   # scf:299-333
    return bool(abs(delta_E) < SCF_conv["delta_E"] and abs(max_DP) < SCF_conv["max_DP"] and abs(RMS_DP) < SCF_conv["RMS_DP"]
                and abs(commutator) < SCF_conv["commutator"])


def _one_electron_traces(integrals, P):
    """tr(P T), tr(P V_NE), tr(P F_field), tr(P G_field) -- all four matrices are symmetric, so tr(P M) = sum(P * M)."""
    return tuple(float(np.sum(P * M)) for M in (integrals.T, integrals.V_NE, integrals.F, integrals.G))


def _energy_tuple(e_kin, e_ne, e_coul, e_exch, e_field, e_fgrad):
    # the reference's component order (scf:400-404): T, V_NE, J, K, correlation (0 here: Hartree-Fock part), field, field gradient
    parts = (e_kin, e_ne, e_coul, e_exch, 0, e_field, e_fgrad)
    return sum(parts), parts


def calculate_restricted_electronic_energy(integrals, P, J, K, calculation, density=None, weights=None, e_X=None, e_C=None):
    """scf:344-404, Hartree-Fock part (the grid terms are added by the Kohn-Sham path): E = tr(P h) + tr(P J)/2 - HFX tr(P K)/4."""
    e_kin, e_ne, e_field, e_fgrad = _one_electron_traces(integrals, P)
    e_coul = 0.5 * float(np.sum(P * J))
    e_exch = -0.25 * calculation.HFX_prop * float(np.sum(P * K))
    return _energy_tuple(e_kin, e_ne, e_coul, e_exch, e_field, e_fgrad)


def construct_restricted_Fock_matrix(integrals, P, HFX_prop, V_XC=None):             # scf:497-531
    V_XC = V_XC if V_XC is not None else 0
    J = calculate_coulomb_matrix(P, integrals.ERI_AO)
    K = calculate_exchange_matrix(P, integrals.ERI_AO)
    F = integrals.T + integrals.V_NE + integrals.F + integrals.G + J - (1 / 2) * K * HFX_prop + V_XC
    return symmetrise(F), J, K


def construct_unrestricted_Fock_matrices(integrals, P_alpha, P_beta, HFX_prop, V_XC_alpha=None, V_XC_beta=None):   # scf:542-589
    """Both spin densities go through the tensor in ONE fused device pass (tf_fock_jk, n_dens = 2)."""
    V_XC_alpha = V_XC_alpha if V_XC_alpha is not None else 0
    V_XC_beta = V_XC_beta if V_XC_beta is not None else 0
    J, K = _device(integrals.ERI_AO).jk(np.stack([P_alpha, P_beta]))
    J_alpha, J_beta, K_alpha, K_beta = J[0], J[1], K[0], K[1]
    F_alpha = integrals.T + integrals.V_NE + J_alpha + J_beta + integrals.F + integrals.G - K_alpha * HFX_prop + V_XC_alpha
    F_beta = integrals.T + integrals.V_NE + J_alpha + J_beta + integrals.F + integrals.G - K_beta * HFX_prop + V_XC_beta
    return symmetrise(F_alpha), symmetrise(F_beta), J_alpha, J_beta, K_alpha, K_beta


def calculate_unrestricted_electronic_energy(integrals, P_alpha, P_beta, J_alpha, J_beta, K_alpha, K_beta, calculation):   # scf:415-486
    """One-electron and Coulomb terms on the total density, exchange per spin: -HFX (tr(Pa Ka) + tr(Pb Kb)) / 2."""
    P_total = P_alpha + P_beta
    e_kin, e_ne, e_field, e_fgrad = _one_electron_traces(integrals, P_total)
    e_coul = 0.5 * float(np.sum(P_total * (J_alpha + J_beta)))
    e_exch = -0.5 * calculation.HFX_prop * (float(np.sum(P_alpha * K_alpha)) + float(np.sum(P_beta * K_beta)))
    return _energy_tuple(e_kin, e_ne, e_coul, e_exch, e_field, e_fgrad)


def format_output_line(E_total, delta_E, max_DP, RMS_DP, damping_factor, step, commutator):   # scf:83-107
    damping = f"{damping_factor:.3f}" if damping_factor != 0 else " ---"
    return f"  {step:3.0f}  {E_total:16.10f}  {delta_E:16.10f} {RMS_DP:16.10f} {max_DP:16.10f} {commutator:16.10f}     {damping}"


SCF_TABLE_HEADER = "  Step          E                 DE             RMS(DP)          MAX(DP)           Error       Damping"   # scf:1328
BIG_SPACER = " " + "~" * 104                               # log_big_spacer, tuna_util.py:1101-1119
SPACER = " " + "~" * 51                                    # log_spacer, tuna_util.py:1072-1090


def log_cycle_header(calculation, o, log):
    """What the reference prints before the first iteration (scf:1313-1329): criteria, convergence acceleration
    (log_convergence_acceleration scf:118-166), title and column header between big spacers."""
    log(" Beginning self-consistent field cycle...\n")
    log(f" Using \"{o['conv']['name']}\" SCF convergence criteria.")
    damping, static = o["damping"] != "none", o["damping"] == "static"
    if o["diis"]:
        tail = (", with static damping." if static else ", with dynamic damping.") if damping else "."
        log(f" Using DIIS, storing {o['max_diis']} matrices, for convergence acceleration" + tail)
    elif damping:
        log(" Using static damping for convergence acceleration." if static else " Using dynamic damping for convergence acceleration.")
    log("")
    log(BIG_SPACER)
    log("                                   Self-consistent Field Cycle Iterations")
    log(BIG_SPACER)
    log(SCF_TABLE_HEADER)
    log(BIG_SPACER)


def log_cycle_rows(r, log):
    """One line per iteration (format_output_line scf:83-107), then the closing spacer and message of check_convergence (scf:324-330)."""
    for row in r["table"]:
        log(format_output_line(row[1], row[2], row[4], row[3], row[6], row[0], row[5]))
    log(BIG_SPACER)
    log(f"\n Self-consistent field converged in {r['n_iter']} cycles!\n")


@dataclass
class Integrals:                                           # tuna_util.py:153-194
    S: np.ndarray
    T: np.ndarray
    V_NE: np.ndarray
    D: np.ndarray
    Q: np.ndarray
    ERI_AO: DeviceERI
    F: np.ndarray | None = None
    G: np.ndarray | None = None

    def __post_init__(self):
        if self.F is None:
            self.F = np.zeros_like(self.S)
        if self.G is None:
            self.G = np.zeros_like(self.S)

    @property
    def H_core(self):
        return self.T + self.V_NE + self.F

    @property
    def one_electron_integrals(self):
        return self.S, self.T, self.V_NE, self.D

    @property
    def n_basis(self):
        return self.S.shape[0]


@dataclass
class Output:                                              # tuna_util.py:205-288 (fields filled by the RHF path)
    energy: float
    kinetic_energy: float
    nuclear_electron_energy: float
    coulomb_energy: float
    exchange_energy: float
    correlation_energy: float
    electric_field_energy: float
    electric_field_gradient_energy: float
    P: np.ndarray
    P_alpha: np.ndarray
    P_beta: np.ndarray
    S: np.ndarray
    X: np.ndarray
    molecular_orbitals: np.ndarray
    molecular_orbitals_alpha: np.ndarray
    molecular_orbitals_beta: np.ndarray
    epsilons: np.ndarray
    epsilons_alpha: np.ndarray
    epsilons_beta: np.ndarray
    density: None
    alpha_density: None
    beta_density: None
    F_alpha: np.ndarray
    F_beta: np.ndarray
    T: np.ndarray
    V_NE: np.ndarray
    integrals: Integrals
    dispersion_energy: float = 0
    n_iterations: int = 0
    table: np.ndarray = field(default_factory=lambda: np.zeros((0, 7)))
    timings: dict = field(default_factory=dict)


def _opts(calculation):
    damping = "none"
    factor = 0.0
    if getattr(calculation, "damping", True):
        if getattr(calculation, "damping_factor", None) is not None:
            damping, factor = "static", float(calculation.damping_factor)
        else:
            damping = "dynamic"
    return dict(conv=calculation.SCF_conv, max_iter=calculation.max_iter, diis=bool(calculation.DIIS),
                max_diis=calculation.max_DIIS_matrices, damping=damping, damping_factor=factor,
                max_damping=calculation.max_damping, hfx=calculation.HFX_prop)


def run_self_consistent_field_cycle(molecule, calculation, integrals: Integrals, V_NN, X, guess_objects, grid_container=None,
                                    silent=True, log=print) -> Output:
    """scf:1292-1435 for the restricted reference.  `molecule` needs n_doubly_occ and partition_ranges; `calculation` the
    fields of `tuna_amd.energy.Calculation` (same names as the reference's Calculation)."""
    if getattr(calculation, "reference", "RHF") == "UHF":
        return _run_unrestricted(molecule, calculation, integrals, V_NN, X, guess_objects, silent, log)
    P, _, _, E = guess_objects
    eri = _device(integrals.ERI_AO)
    eng = eri.engine
    o = _opts(calculation)
    if not silent:
        log_cycle_header(calculation, o, log)
    Fext = integrals.F + integrals.G
    import os
    if eng.world > 1 and not getattr(eng, "has_allreduce", False) and not os.environ.get("TUNA_AMD_HOST_SCF"):
        from . import distributed as tdist
        tdist.attach_allreduce(eng)                          # sharded tensor: the native cycle all-reduces the partial [J;K] per build
    if eng.world == 1 or not os.environ.get("TUNA_AMD_HOST_SCF"):
        try:
            r = eng.scf_rhf(integrals.S, integrals.T, integrals.V_NE, P, E, molecule.n_doubly_occ, V_NN, X=X,
                            Fext=Fext if np.any(Fext) else None, n_atom_ao=molecule.partition_ranges, **o)
        except TunaError as e:
            if not silent and getattr(e, "partial", None) is not None:
                for row in e.partial["table"]:
                    log(format_output_line(row[1], row[2], row[4], row[3], row[6], row[0], row[5]))
            raise
    else:
        r = _python_level_cycle(molecule, calculation, integrals, V_NN, X, P, E, o)
    if not silent:
        log_cycle_rows(r, log)
    c = r["components"]
    Pf, Cm, eps, F = r["P"], r["C"], r["epsilons"], r["F"]
    return Output(r["energy"], c[0], c[1], c[2], c[3], c[4], c[5], c[6], Pf, Pf / 2, Pf / 2, integrals.S, X, Cm, Cm, Cm, eps, eps, eps,
                  None, None, None, F / 2, F / 2, integrals.T, integrals.V_NE, integrals, 0, r["n_iter"], r["table"],
                  {k: r[k] for k in ("fock_seconds", "eig_seconds", "wall_seconds") if k in r})


def _python_level_cycle(molecule, calculation, integrals, V_NN, X, P, E, o):
    """The iteration order of scf:1072-1154 with device Fock builds (sharded tensor + all-reduce) and rocSOLVER
    diagonalisations, orchestrated from the host: the cross-check of the native cycle (TUNA_AMD_HOST_SCF=1 selects it for a
    sharded tensor).  Hartree-Fock only.  Every rank runs it redundantly."""
    if getattr(calculation, "DFT_calculation", False):
        raise TunaError("the host-orchestrated cycle is Hartree-Fock only: Kohn-Sham runs in the native cycle (tf_scf_rhf)")
    gen = _cycle_steps(molecule, integrals, V_NN, X, P, E, o, integrals.F + integrals.G)
    try:
        D = next(gen)
        while True:
            D = gen.send(integrals.ERI_AO.jk(D))
    except StopIteration as done:
        return done.value


def run_cycles_in_lockstep(molecule, calculation, integrals, V_NN, X, guess_objects, external_terms):
    """Several restricted Hartree-Fock cycles on the SAME tensor, one per external one-electron term (the finite-field evaluations of
    tuna_energy.py:355-396, 483-540 differ in nothing but F_fld), advanced together: every iteration builds the Fock matrices of all
    cycles still running in ONE call -- their densities go through the tensor as a batch (n_dens = cycles: pairs of densities per
    pass) instead of one pass each.  Each cycle follows exactly the iteration order of a run on its own (scf:1072-1154) and stops by
    its own criteria.  Returns the result dictionaries in the order of `external_terms`."""
    if getattr(calculation, "DFT_calculation", False) or getattr(calculation, "reference", "RHF") == "UHF":
        raise TunaError("finite-field batches run restricted Hartree-Fock cycles in this build")
    P, _, _, E = guess_objects
    o = _opts(calculation)
    gens = [_cycle_steps(molecule, integrals, V_NN, X, P, E, o, integrals.G + term) for term in external_terms]
    wanted = {k: next(g) for k, g in enumerate(gens)}        # cycle -> the density it needs J and K for
    results = [None] * len(gens)
    while wanted:
        keys = sorted(wanted)
        J, K = integrals.ERI_AO.jk(np.stack([wanted[k] for k in keys]))
        for n, k in enumerate(keys):
            try:
                wanted[k] = gens[k].send((J[n], K[n]))
            except StopIteration as done:
                results[k] = done.value
                del wanted[k]
    return results


def run_cycles_in_native_lockstep(molecule, calculation, integrals, V_NN, X, guess_objects, external_terms):
    """The same batch inside the library (tf_scf_rhf_batch): every cycle is the native cycle (tf_scf.hip.h: run_rhf, on a host thread
    and a workspace of its own -- DIIS history, eigenvector refinement state, damping), and the cycles meet once per iteration in the
    Fock build, where the densities of all cycles still iterating go through the tensor together."""
    if getattr(calculation, "DFT_calculation", False) or getattr(calculation, "reference", "RHF") == "UHF":
        raise TunaError("finite-field batches run restricted Hartree-Fock cycles in this build")
    P, _, _, E = guess_objects
    eng = _device(integrals.ERI_AO).engine
    o = _opts(calculation)
    n = len(external_terms)
    res = eng.scf_rhf_batch(integrals.S, integrals.T, integrals.V_NE, [P] * n, [E] * n, molecule.n_doubly_occ, V_NN, X=X,
                            Fexts=[integrals.G + t for t in external_terms], n_atom_ao=molecule.partition_ranges, **o)
    integrals.ERI_AO.n_builds += res[0]["tensor_passes"] if res else 0
    return res


def _cycle_steps(molecule, integrals, V_NN, X, P, E, o, Fext):
    """One restricted Hartree-Fock cycle as a generator: yields the density whose J and K it needs next, receives (J, K), returns
    the result dictionary (scf:1072-1154; DIIS scf:960-1061, damping scf:1110-1154)."""
    import time
    S, T, V = integrals.S, integrals.T, integrals.V_NE
    eng = integrals.ERI_AO.engine
    thr = o["conv"]
    n_occ = molecule.n_doubly_occ
    ranges = list(molecule.partition_ranges)
    P_old = np.zeros_like(P)
    P_zero = np.zeros_like(P)                      # "P_old_before_damping" is always zero in the reference (scf:1154/1373)
    Fock_vector, err_vector, table = [], [], []
    t0 = time.perf_counter()

    def pops(D):
        d = np.einsum("ij,ji->i", D, S)
        return np.array([d[:ranges[0]].sum(), d[ranges[0]:].sum() if len(ranges) > 1 else 0.0])

    for step in range(1, o["max_iter"] + 1):
        E_old, P_very_old, P_old = E, P_old, P
        J, K = yield P
        F = symmetrise(T + V + Fext + J - (1 / 2) * K * o["hfx"])
        e = X.T @ (F @ P @ S - S @ P @ F) @ X
        commutator = np.mean(e * e) ** (1 / 2)
        err_vector.append(np.concatenate((e.flatten(), e.flatten())))
        Fock_vector.append(F)
        if len(Fock_vector) > o["max_diis"]:
            del Fock_vector[0], err_vector[0]
        eps, C = eng.diagonalise(F, X)
        P = construct_density_matrix(C, n_occ, 2)
        comps = (np.sum(P * T), np.sum(P * V), (1 / 2) * np.sum(P * J), -(1 / 4) * np.sum(P * K) * o["hfx"], 0.0, np.sum(P * Fext), 0.0)
        E = sum(comps)
        if step > 2 and o["diis"] and commutator < 0.3:
            n = len(err_vector)
            errs = np.array(err_vector)
            B = np.empty((n + 1, n + 1))
            B[:n, :n] = errs @ errs.T
            B[:n, -1] = -1
            B[-1, :n] = -1
            B[-1, -1] = 0
            rhs = np.zeros(n + 1)
            rhs[-1] = -1
            try:
                coeffs = np.linalg.solve(B, rhs)[:n]
                _, C_d = eng.diagonalise(np.tensordot(coeffs, np.array(Fock_vector), axes=(0, 0)), X)
                P = symmetrise(construct_density_matrix(C_d, n_occ, 2))
            except np.linalg.LinAlgError:
                Fock_vector.clear()
                err_vector.clear()
        P_before = P
        factor = 0.0
        if o["damping"] == "static":
            factor = o["damping_factor"]
        elif o["damping"] == "dynamic" and commutator > 0.01 and step > 1:
            A_out, A1_in, A1_out, A2_in = pops(P_before), pops(P_old), pops(P_zero), pops(P_very_old)
            den = A_out - A1_out - A1_in + A2_in
            alpha = (A_out - A1_out) / den if den.all() != 0 else [0, 0]
            factor = ((alpha[0] * ranges[0] + alpha[1] * ranges[1]) / (ranges[0] + ranges[1])) if len(ranges) == 2 else alpha[0] * ranges[0]
            factor = max(factor, 0)
            factor = factor if factor < min(o["max_damping"], 1) else o["max_damping"]
        P = factor * P_old + (1 - factor) * P_before
        dE, maxDP, rmsDP = calculate_SCF_changes(E, E_old, P, P_old)
        table.append([step, E + V_NN, dE, rmsDP, maxDP, commutator, float(factor)])
        if check_convergence(thr, step, dE, maxDP, rmsDP, commutator):
            return dict(energy=E + V_NN, components=np.array(comps, dtype=float), P=P, C=C, epsilons=eps, F=F, table=np.array(table),
                        n_iter=step, converged=True, wall_seconds=time.perf_counter() - t0)
    raise TunaError(f"Self-consistent field not converged in {o['max_iter']} iterations! Increase maximum iterations or give up.", -4)


def _run_unrestricted(molecule, calculation, integrals, V_NN, X, guess_objects, silent, log) -> Output:
    """run_unrestricted_SCF_cycle (scf:1165-1281) inside the outer loop (scf:1292-1435).  One GPU: the whole cycle in the library
    (tf_scf_uhf); a sharded tensor (world > 1) or TUNA_AMD_HOST_UHF=1: the host-orchestrated loop below."""
    import os
    eng = _device(integrals.ERI_AO).engine
    if os.environ.get("TUNA_AMD_HOST_UHF"):
        return _python_level_unrestricted(molecule, calculation, integrals, V_NN, X, guess_objects, silent, log)
    if eng.world > 1 and not getattr(eng, "has_allreduce", False):
        from . import distributed as tdist
        tdist.attach_allreduce(eng)
    _, Pa, Pb, E = guess_objects
    o = _opts(calculation)
    if not silent:
        log_cycle_header(calculation, o, log)
    Fext = integrals.F + integrals.G
    try:
        r = eng.scf_uhf(integrals.S, integrals.T, integrals.V_NE, Pa, Pb, E, molecule.n_alpha, molecule.n_beta, V_NN, X=X,
                        Fext=Fext if np.any(Fext) else None, n_atom_ao=molecule.partition_ranges, **o)
    except TunaError as e:
        if not silent and getattr(e, "partial", None) is not None:
            for row in e.partial["table"]:
                log(format_output_line(row[1], row[2], row[4], row[3], row[6], row[0], row[5]))
        raise
    if not silent:
        log_cycle_rows(r, log)
    c = r["components"]
    (Pa, Pb), (Ca, Cb), (Fa, Fb), (eps_a, eps_b) = r["P_spin"], r["C_spin"], r["F_spin"], r["epsilons_spin"]
    eps = np.concatenate((eps_a, eps_b))
    order = np.argsort(eps)
    C_all = np.concatenate((Ca, Cb), axis=1)[:, order]
    return Output(r["energy"], c[0], c[1], c[2], c[3], c[4], c[5], c[6], r["P"], Pa, Pb, integrals.S, X, C_all, Ca, Cb, eps[order], eps_a, eps_b,
                  None, None, None, Fa, Fb, integrals.T, integrals.V_NE, integrals, 0, r["n_iter"], r["table"],
                  {k: r[k] for k in ("fock_seconds", "eig_seconds", "wall_seconds")})


def _python_level_unrestricted(molecule, calculation, integrals, V_NN, X, guess_objects, silent, log) -> Output:
    """The same cycle with fused two-density device Fock builds (all-reduced over ranks when the tensor is sharded) and device
    diagonalisations; O(N^2) bookkeeping on the host as in the reference.  Reference quirks kept: "P_very_old" and
    "P_old_before_damping" of each spin are zero matrices in every iteration (scf:1281 vs scf:1394)."""
    import time
    _, Pa, Pb, E = guess_objects
    S = integrals.S
    eng = _device(integrals.ERI_AO).engine
    o = _opts(calculation)
    thr = o["conv"]
    n_alpha, n_beta = molecule.n_alpha, molecule.n_beta
    ranges = list(molecule.partition_ranges)
    P = Pa + Pb
    Fock_vector, err_vector, table = [], [], []
    t0 = time.perf_counter()

    def pops(D):
        d = np.einsum("ij,ji->i", D, S)
        return np.array([d[:ranges[0]].sum(), d[ranges[0]:].sum() if len(ranges) > 1 else 0.0])

    def commutator_of(F, D):
        e = X.T @ (F @ D @ S - S @ D @ F) @ X
        return np.mean(e * e) ** (1 / 2), e

    def damp(P_new, P_old_spin, commutator_spin, step):
        factor = 0.0
        if o["damping"] == "static":
            factor = o["damping_factor"]
        elif o["damping"] == "dynamic" and commutator_spin > 0.01 and step > 1:
            A_out, A1_in = pops(P_new), pops(P_old_spin)
            den = A_out - A1_in                                  # the other two populations are of zero matrices
            alpha = A_out / den if den.all() != 0 else [0, 0]
            factor = ((alpha[0] * ranges[0] + alpha[1] * ranges[1]) / (ranges[0] + ranges[1])) if len(ranges) == 2 else alpha[0] * ranges[0]
            factor = max(factor, 0)
            factor = factor if factor < min(o["max_damping"], 1) else o["max_damping"]
        return factor * P_old_spin + (1 - factor) * P_new, factor

    if not silent:
        log_cycle_header(calculation, o, log)
    for step in range(1, o["max_iter"] + 1):
        E_old, P_old, Pa_old, Pb_old = E, P, Pa, Pb
        Fa, Fb, Ja, Jb, Ka, Kb = construct_unrestricted_Fock_matrices(integrals, Pa, Pb, o["hfx"])
        ca, ea = commutator_of(Fa, Pa)
        cb, eb = commutator_of(Fb, Pb)
        commutator = max(ca, cb)
        err_vector.append(np.concatenate((ea.flatten(), eb.flatten())))
        Fock_vector.append((Fa, Fb))
        if len(Fock_vector) > o["max_diis"]:
            del Fock_vector[0], err_vector[0]
        eps_a, Ca = eng.diagonalise(Fa, X)
        eps_b, Cb = eng.diagonalise(Fb, X)
        Pa, Pb = construct_density_matrix(Ca, n_alpha, 1), construct_density_matrix(Cb, n_beta, 1)
        E, comps = calculate_unrestricted_electronic_energy(integrals, Pa, Pb, Ja, Jb, Ka, Kb, calculation)
        if step > 2 and o["diis"] and commutator < 0.3:
            n = len(err_vector)
            errs = np.array(err_vector)
            B = np.empty((n + 1, n + 1))
            B[:n, :n] = errs @ errs.T
            B[:n, -1] = -1
            B[-1, :n] = -1
            B[-1, -1] = 0
            rhs = np.zeros(n + 1)
            rhs[-1] = -1
            try:
                coeffs = np.linalg.solve(B, rhs)[:n]
                _, Ca_d = eng.diagonalise(np.tensordot(coeffs, np.array([f[0] for f in Fock_vector]), axes=(0, 0)), X)
                _, Cb_d = eng.diagonalise(np.tensordot(coeffs, np.array([f[1] for f in Fock_vector]), axes=(0, 0)), X)
                Pa, Pb = symmetrise(construct_density_matrix(Ca_d, n_alpha, 1)), symmetrise(construct_density_matrix(Cb_d, n_beta, 1))
            except np.linalg.LinAlgError:
                Fock_vector.clear()
                err_vector.clear()
        Pa, fa = damp(Pa, Pa_old, ca, step)
        Pb, fb = damp(Pb, Pb_old, cb, step)
        P = Pa + Pb
        dE, maxDP, rmsDP = calculate_SCF_changes(E, E_old, P, P_old)
        table.append([step, E + V_NN, dE, rmsDP, maxDP, commutator, float(max(fa, fb))])
        if not silent:
            log(format_output_line(E + V_NN, dE, maxDP, rmsDP, max(fa, fb), step, commutator))
        if check_convergence(thr, step, dE, maxDP, rmsDP, commutator):
            eps = np.concatenate((eps_a, eps_b))
            order = np.argsort(eps)
            C_all = np.concatenate((Ca, Cb), axis=1)[:, order]
            if not silent:
                log(BIG_SPACER)
                log(f"\n Self-consistent field converged in {step} cycles!\n")
            return Output(E + V_NN, comps[0], comps[1], comps[2], comps[3], comps[4], comps[5], comps[6], P, Pa, Pb, S, X, C_all, Ca, Cb,
                          eps[order], eps_a, eps_b, None, None, None, Fa, Fb, integrals.T, integrals.V_NE, integrals, 0, step,
                          np.array(table), {"wall_seconds": time.perf_counter() - t0})
    raise TunaError(f"Self-consistent field not converged in {o['max_iter']} iterations! Increase maximum iterations or give up.", -4)
