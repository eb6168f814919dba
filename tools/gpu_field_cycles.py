"""GPU: the eight finite-field cycles of a polarisability at the bench workload, one by one on one engine: iterations and energy per field
(a bisecting aid: TF_JK_CLASS_DIAGONAL=0 / TF_EIGH_BLOCKS=0 switch the round-4 paths off).  usage: python tools/gpu_field_cycles.py [N=400]"""
import os, sys, time, types
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench
from oracle import scf_oracle as so
from tuna_amd import molecule as mol, properties as props
from tuna_amd._lib import TunaError
from tuna_amd.engine import Engine

wl = "synth-" + (sys.argv[1] if len(sys.argv) > 1 else "400")
atoms, shells, aos, nocc, desc = bench.build_workload(wl)
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    S, T, V, D, Q = eng.one_electron(xyz, chg, [0.0, 0.0, 0.5 * atoms[-1].origin[2]], spherical=True)
    X, _, _ = eng.orthogonaliser(S)
    P0, E0 = so.core_guess(T, V, X, nocc)
    nao = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
    h = props.SECOND_ELEC_DERIVATIVE_STEP
    fields = [[0, 0, 2 * h], [0, 0, h], [0, 0, -h], [0, 0, -2 * h], [2 * h, 0, 0], [h, 0, 0], [-h, 0, 0], [-2 * h, 0, 0]]
    for rep in range(2):
        for f in fields:
            Fext = f[0] * D[0] + f[1] * D[1] + f[2] * D[2]
            t0 = time.perf_counter()
            try:
                r = eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, Fext=Fext, conv="tight", damping="dynamic", n_atom_ao=nao, max_iter=100)
                print(f"rep {rep} field {f}: {r['n_iter']} iterations, {1e3 * (time.perf_counter() - t0):.1f} ms, E = {r['energy']:.10f}", eng.jk_path_stats(), eng.eigh_stats(), flush=True)
            except TunaError as e:
                print(f"rep {rep} field {f}: {e}", eng.jk_path_stats(), eng.eigh_stats(), flush=True)
