"""GPU: the wide Fock pass of the tiles layout (4 / 8 densities per pass: densities = columns of the matrix-core B operands) against the
one-density pass, and its timing.  usage: python tools/gpu_tiles_nd.py [workload] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from bench import build_workload  # noqa: E402
from tuna_amd.engine import Engine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "synth-400"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
atoms, shells, aos, nocc, desc = build_workload(wl)
eng = Engine(0)
eng.set_basis(aos)
eng.build_eri(True, layout="tiles")
N = eng.N
rng = np.random.default_rng(3)
dens = []
for d in range(8):
    A = rng.standard_normal((N, N))
    dens.append(A + A.T)
dev = torch.device("cuda", 0)
dP = torch.from_numpy(np.stack(dens)).to(dev)
stream = torch.cuda.current_stream().cuda_stream
ref = torch.zeros((2, 8, N, N), dtype=torch.float64, device=dev)
for d in range(8):
    eng.fock_jk_device(dP[d].data_ptr(), ref[0, d].data_ptr(), ref[1, d].data_ptr(), 1, stream)
torch.cuda.synchronize()
refh = ref.cpu().numpy()
for nd in (2, 3, 4, 5, 8):
    out = torch.zeros((2, nd, N, N), dtype=torch.float64, device=dev)
    eng.fock_jk_device(dP.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), nd, stream)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    eJ = max(np.abs(o[0, d] - refh[0, d]).max() / np.abs(refh[0, d]).max() for d in range(nd))
    eK = max(np.abs(o[1, d] - refh[1, d]).max() / np.abs(refh[1, d]).max() for d in range(nd))
    for _ in range(2):
        eng.fock_jk_device(dP.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), nd, stream)
    torch.cuda.synchronize()
    eng.jk_profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.fock_jk_device(dP.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), nd, stream)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    ks, kn = eng.jk_profile_read()
    eng.jk_profile(False)
    print(f"{wl} nd={nd}: rel err J {eJ:.2e} K {eK:.2e} | pass {1e3 * el:.3f} ms = {nd / el:.0f} Fock matrices/s (tile kernel {1e3 * ks / max(kn, 1):.3f} ms per launch, {kn // steps} launches)", flush=True)
