"""GPU (one card): what each rank of a G-way sharded 400-AO Fock build would spend in its local J/K pass -- every rank's context is
created in turn on cuda:0, its share of the tensor built and its partial build timed.  The slowest rank bounds the parallel step.
usage: python tools/gpu_shard_timing.py [N=400] [worlds=1,2,4,8]"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()                                                    # (before the library touches the device)
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
            eng.jk_profile(False)
            # the whole local build on device-resident buffers (pack, J/K kernel, reductions, final): what a rank spends before the exchange
            dev = torch.device("cuda", 0)
            dP = torch.from_numpy(P).to(dev); dJK = torch.zeros((2, eng.N, eng.N), dtype=torch.float64, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(3):
                eng.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), 1, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                eng.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), 1, st)
            e1.record(); torch.cuda.synchronize()
            dev_build = e0.elapsed_time(e1) / 20 * 1e-3
            times.append((ksec / n, wall, dev_build)); gbs.append(eng.eri_storage()["bytes"] / 1e9)
    k = [t[0] for t in times]; w = [t[1] for t in times]; b = [t[2] for t in times]
    print(f"world {world}: local build on device buffers (pack + J/K kernel + reductions + final) per rank min {1e3*min(b):.3f} max {1e3*max(b):.3f} ms", flush=True)
    print(f"world {world}: J/K kernel per rank min {1e3*min(k):.2f} max {1e3*max(k):.2f} ms (ideal {1e3*sum(k)/world:.2f} at perfect balance); "
          f"host-buffer build wall max {1e3*max(w):.2f} ms; stored GB per rank {min(gbs):.2f}-{max(gbs):.2f}; ERI build max {max(eri):.3f} s", flush=True)
