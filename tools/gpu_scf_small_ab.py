"""GPU: A/B of the N2/cc-pVTZ SCF leg (BASELINE configs[1]) between library variants; prints warm SCF wall, eigensolver time and the
refinement statistics (TF_DEBUG lines).  usage: python tools/gpu_scf_small_ab.py name1 name2 ..."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r"""
import sys, time, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import bench
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
atoms, shells, aos, nocc, desc = bench.build_workload("n2-cc-pvtz")
xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
with Engine(0) as eng:
    for rep in range(3):
        eng.set_basis(aos).build_eri(True)
        t1 = time.perf_counter()
        S, T, V, _, _ = eng.one_electron(xyz, chg, [0, 0, 0.5 * atoms[1].origin[2]])
        X, _, _ = eng.orthogonaliser(S)
        _, C0 = eng.diagonalise(T + V, X)
        P0 = 2.0 * C0[:, :nocc] @ C0[:, :nocc].T
        P0 = 0.5 * (P0 + P0.T)
        E0 = float(np.sum(P0 * (T + V)))
        r = eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="none", n_atom_ao=[30, 30])
        t_scf = time.perf_counter() - t1
print(json.dumps({"energy_Eh": r["energy"], "iterations": r["n_iter"], "scf_wall_ms": 1e3 * t_scf, "eigensolver_ms": 1e3 * r["eig_seconds"],
                  "fock_kernel_ms": 1e3 * r["fock_seconds"], "native_cycle_ms": 1e3 * r["wall_seconds"]}))
"""
for rep in range(2):
    for name in sys.argv[1:] or ["base"]:
        env = dict(os.environ, TF_DEBUG="1")                  # name = a library variant, or VAR=value for the base library with that variable
        if "=" in name:
            k, v = name.split("=", 1)
            env[k] = v
        env["TUNAFOCK_LIB"] = os.path.join(ROOT, "tuna_amd", "libtunafock.so" if (name == "base" or "=" in name) else f"libtunafock_{name}.so")
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True)
        steps = [int(l.split("after")[1].split()[0]) for l in out.stderr.splitlines() if "tf refine/lds" in l and "after" in l]
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
        print(name, rep, line, "refine solves", len(steps), "mean steps %.2f" % (sum(steps) / max(1, len(steps))), flush=True)
