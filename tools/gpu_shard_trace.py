"""GPU (one card): kernel durations of one rank's partial Fock build in a G-way sharded run (under rocprofv3 --kernel-trace --stats).
usage: python tools/gpu_shard_trace.py WORLD [RANK]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
world = int(sys.argv[1]); rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
counts = mol.synthetic_counts(400)
atoms = mol.make_atoms(["AR", "AR"], 7.1)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
rng = np.random.default_rng(0)
with Engine(0, rank, world) as eng:
    eng.set_basis(aos).build_eri(True)
    A = rng.standard_normal((eng.N, eng.N)); P = A + A.T
    for _ in range(12):
        eng.fock_jk(P)
