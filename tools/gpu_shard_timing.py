"""GPU (one card): what each rank of a G-way sharded 400-AO Fock build would spend in its local J/K pass -- every rank's context is
created in turn on cuda:0, its share of the tensor built and its partial build timed.  The slowest rank bounds the parallel step.
usage: python tools/gpu_shard_timing.py [N=400] [worlds=1,2,4,8]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
worlds = [int(w) for w in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8").split(",")]
counts = mol.synthetic_counts(N)
atoms = mol.make_atoms(["AR", "AR"], 7.1)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
rng = np.random.default_rng(0)
for world in worlds:
    times, gbs, eri = [], [], []
    for rank in range(world):
        with Engine(0, rank, world) as eng:
            t0 = time.perf_counter(); eng.set_basis(aos).build_eri(True); eri.append(time.perf_counter() - t0)
            A = rng.standard_normal((eng.N, eng.N)); P = A + A.T
            eng.fock_jk(P)
            eng.jk_profile(True)
            t0 = time.perf_counter()
            for _ in range(5):
                eng.fock_jk(P)
            wall = (time.perf_counter() - t0) / 5
            ksec, n = eng.jk_profile_read()
            times.append((ksec / n, wall)); gbs.append(eng.eri_storage()["bytes"] / 1e9)
    k = [t[0] for t in times]; w = [t[1] for t in times]
    print(f"world {world}: J/K kernel per rank min {1e3*min(k):.2f} max {1e3*max(k):.2f} ms (ideal {1e3*sum(k)/world:.2f} at perfect balance); "
          f"host-buffer build wall max {1e3*max(w):.2f} ms; stored GB per rank {min(gbs):.2f}-{max(gbs):.2f}; ERI build max {max(eri):.3f} s", flush=True)
