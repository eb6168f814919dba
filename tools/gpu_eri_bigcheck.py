"""GPU: the tensor of a synthetic workload larger than the test suite builds (default N = 600), pinned to the oracle on random
shell-quartet blocks (the helper of tests/test_gpu_parity.py) + 8-fold symmetry of samples.  usage: python tools/gpu_eri_bigcheck.py [N] [blocks]"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as tp
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 120
counts = mol.synthetic_counts(N)
atoms = mol.make_atoms(["AR", "AR"], 7.1)
shells = mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)})
with Engine(0) as eng:
    eng.set_basis(mol.expand_cartesian_aos(shells))
    for rep in range(2):
        t0 = time.perf_counter(); eng.build_eri(True); wall = time.perf_counter() - t0
    t = eng.eri_timings()
    print(f"N = {eng.N}: build {wall*1e3:.1f} ms wall, ERI kernels {t['cart_kernel_s']*1e3:.1f} ms, device {t['total_s']*1e3:.1f} ms, stored {eng.eri_storage()['bytes']/1e9:.2f} GB")
    worst, n = tp._shell_quartet_blocks_against_oracle(eng, shells, nblk, 5, True)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, eng.N, size=(4000, 4)).astype(np.int32)
    v = eng.sample_eri(idx)
    sym = max(np.abs(v - eng.sample_eri(idx[:, p])).max() for p in [(1, 0, 2, 3), (0, 1, 3, 2), (2, 3, 0, 1), (3, 2, 0, 1)])
    print(f"{nblk} shell-quartet blocks, {n} elements: max |GPU - oracle| = {worst:.2e}; 8-fold symmetry of 4000 samples: {sym:.2e}")
    assert worst < 1e-12 and sym < 1e-12
