#!/usr/bin/env python3
"""Golden PRINTED TEXT of the reference for the BASELINE input lines (build container only; needs /root/reference and oracle/_ref).

What the reference prints around the SCF of a single-point run is produced here by the reference's OWN code executed from its
source text, with a `log` that records instead of printing:
  * the block of run_self_consistent_field_cycle (tuna_scf.py:1292-1435): "Beginning self-consistent field cycle...", the criteria
    line (tuna_scf.py:1319: the one statement that does not parse under Python 3.10 -- its message is re-created from the same
    format), log_convergence_acceleration, the table title / header between big spacers, one format_output_line per iteration
    (tuna_scf.py:83-107), the spacer and "converged in N cycles" of check_convergence (tuna_scf.py:299-333);
  * print_SCF_energy (tuna_kernel.py:828-866) and the "Final single point energy" line (tuna_kernel.py:1305), their log() calls
    evaluated from the source;
  * log_spacer / log_big_spacer from tuna_util.py:1072-1119.
Runs are the DEFAULT path: SAD guess, "medium" thresholds, DIIS 6 + dynamic damping (tools/make_golden.py: make_sad_golden).
Only DATA is written: tests/golden/output_text.json = {tag: {"input_line": ..., "lines": [...]}}.
"""
from __future__ import annotations

import ast
import json
import os
import sys
import types

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg  # noqa: E402
from make_golden import REF, ROOT, GOLD, mol, orc  # noqa: E402


def reference_log_helpers(capture):
    """log_spacer / log_big_spacer of tuna_util.py from source text, bound to a recording log."""
    src = open(os.path.join(REF, "TUNA", "tuna_util.py")).read()
    lines, _ = mg._parseable_lines(os.path.join(REF, "TUNA", "tuna_util.py"))
    tree = ast.parse("\n".join(lines))
    ns = {"log": capture}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("log_spacer", "log_big_spacer"):
            node.returns = None
            for a in node.args.args + node.args.kwonlyargs:
                a.annotation = None
            exec(compile(ast.Module([node], []), "tuna_util.py", "exec"), ns)
    return ns["log_spacer"], ns["log_big_spacer"]


def first_log_argument(path, needle, env):
    """Evaluates the message expression of the log(...) call on the source line that contains `needle`."""
    for line in open(path).read().split("\n"):
        if needle in line and "log(" in line:
            call = ast.parse(line.strip()).body[0].value
            return eval(compile(ast.Expression(call.args[0]), path, "eval"), env)
    raise RuntimeError(f"{needle!r} not found in {path}")


def main():
    assert orc.ref_engine() is not None, "run oracle/build_ref.sh first"
    scf = mg.load_reference_scf()
    blocks, ortho = mg.load_reference_kernel_bits()
    rec = []

    def capture(message, calculation=None, priority=1, silent=False, end="\n", colour="light_grey"):
        if silent or priority > 2:
            return
        rec.append((message, end))
    log_spacer, log_big_spacer = reference_log_helpers(capture)
    scf.log, scf.log_spacer, scf.log_big_spacer = capture, log_spacer, log_big_spacer

    # SAD guess exactly as make_sad_golden does
    src = open(os.path.join(REF, "TUNA", "tuna_guess.py")).read()
    tree = ast.parse(src)
    from scipy.linalg import block_diag
    ns = {"np": np, "ndarray": np.ndarray, "block_diag": block_diag, "Atom": object}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in ("project_density_matrix", "form_minimal_basis_superposition_density"):
            exec(compile(ast.Module([node], []), "tuna_guess.py", "exec"), ns)
    dens = {k: (None if v["density"] is None else np.array(v["density"])) for k, v in
            json.load(open(os.path.join(ROOT, "tuna_amd", "data", "atomic_data.json"))).items()}
    ints_ref = orc.ref_engine()
    kernel_py = os.path.join(REF, "TUNA", "tuna_kernel.py")
    out = {}
    cases = {
        "c1_h2_sto3g": ("SPE : H H 0.74 : HF STO-3G", ["H", "H"], mol.angstrom_to_bohr(0.74), "STO-3G", 1),
        "c2_n2_ccpvtz": ("SPE : N N 1.0977 : HF CC-PVTZ", ["N", "N"], mol.angstrom_to_bohr(1.0977), "cc-pVTZ", 7),
        "c3_ar2_ccpvqz": ("SPE : AR AR 3.76 : HF CC-PVQZ", ["AR", "AR"], mol.angstrom_to_bohr(3.76), "cc-pVQZ", 18),
        "n2_ccpvdz": ("SPE : N N 1.0977 : HF CC-PVDZ", ["N", "N"], mol.angstrom_to_bohr(1.0977), "cc-pVDZ", 7),
    }
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    for tag, (line, sym, R, basis, nocc) in cases.items():
        if only and tag not in only:
            continue
        atoms, shells, aos = mg.system(sym, R, basis)
        S, T, V, D, Q, E = mg.one_e_and_eri(atoms, aos)
        U = mg.reference_U(shells, blocks)
        Ss, Ts_, Vs, Es = mg.to_spherical(U, S), mg.to_spherical(U, T), mg.to_spherical(U, V), mg.eri_to_spherical(U, E)
        del E
        X, smallest, S_inv = ortho(Ss, None, True)
        ref_atoms = [types.SimpleNamespace(density=dens[a.symbol]) for a in atoms]
        P_min = ns["form_minimal_basis_superposition_density"](ref_atoms)
        _, _, aos_min = mg.system(sym, R, "STO-3G")
        S_cross = np.asarray(ints_ref.calculate_cross_basis_overlap_matrix(aos.n, aos_min.n, orc.ref_basis_list(aos), orc.ref_basis_list(aos_min), 4))
        P_spin = ns["project_density_matrix"](P_min, S_cross, S_inv, U)
        Pa = P_spin * (nocc / np.trace(P_spin @ Ss))
        P0 = Pa + Pa
        E0 = float(np.einsum("mn,mn->", Ts_ + Vs, P0, optimize=True))
        n_sph = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
        molecule = types.SimpleNamespace(n_doubly_occ=nocc, partition_ranges=n_sph, atoms=atoms, n_electrons=2 * nocc, n_alpha=nocc, n_beta=nocc)
        calc = mg.Calc(mg.CONV["medium"], damping=True)
        rec.clear()
        o = scf.run_self_consistent_field_cycle(molecule, calc, mg.Ints(Ss, Ts_, Vs, Es), mol.nuclear_repulsion(atoms), X,
                                                (P0, Pa, Pa, E0), (None, None, None, None), False)
        # tuna_scf.py:1319 (unparsable here, replaced by `pass`): the same message from the same format, inserted after the first line
        crit = f" Using \"{calc.SCF_conv['name']}\" SCF convergence criteria."
        rec.insert(1, (crit, "\n"))
        env = {"final_energy": o.energy, "np": np}
        rec.append((first_log_argument(kernel_py, "Restricted Hartree-Fock energy:", env), "\n"))
        rec.append((first_log_argument(kernel_py, "Final single point energy:", env), "\n"))
        text = "".join(m + e for m, e in rec)
        out[tag] = {"input_line": line, "lines": text.split("\n"), "energy": o.energy}
        print(tag, "lines", len(out[tag]["lines"]), "E", o.energy)
    path = os.path.join(GOLD, "output_text.json")
    if only and os.path.exists(path):
        old = json.load(open(path))
        old.update(out)
        out = old
    json.dump(out, open(path, "w"), indent=0)


if __name__ == "__main__":
    main()
