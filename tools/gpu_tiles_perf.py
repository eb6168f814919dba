"""GPU: time the Fock build of the tiles layout on a bench workload (kernel time from the in-library HIP events, whole build from the
host clock) -- usage: python tools/gpu_tiles_perf.py [workload] [steps]; knobs through the environment (TF_TILE_KSUB, TF_TILE_PART_STEPS,
TF_ERI_LAYOUT=p for the packed layout as the baseline)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from bench import build_workload  # noqa: E402
from tuna_amd.engine import Engine  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "synth-400"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
atoms, shells, aos, nocc, desc = build_workload(wl)
eng = Engine(0)
eng.set_basis(aos)
t0 = time.perf_counter()
eng.build_eri(True)
torch.cuda.synchronize()
t1 = time.perf_counter()
eng.build_eri(True)
torch.cuda.synchronize()
t2 = time.perf_counter()
st = eng.eri_storage()
N = eng.N
A = np.random.default_rng(0).standard_normal((N, N))
P = A + A.T
dev = torch.device("cuda", 0)
dP = torch.from_numpy(P[None]).to(dev)
dJK = torch.zeros((2, 1, N, N), dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    eng.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), 1, stream)
torch.cuda.synchronize()
eng.jk_profile(True)
t0 = time.perf_counter()
for _ in range(steps):
    eng.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), 1, stream)
torch.cuda.synchronize()
el = time.perf_counter() - t0
ks, kn = eng.jk_profile_read()
eng.jk_profile(False)
J = dJK[0, 0].cpu().numpy()
K = dJK[1, 0].cpu().numpy()
print(f"{wl} layout={st['layout']} ksub={os.environ.get('TF_TILE_KSUB', '64')} bytes={st['bytes'] / 1e9:.3f} GB eri_build cold {t1 - t0 + 0:.3f}s warm {t2 - t1:.3f}s | "
      f"kernel {1e3 * ks / max(kn, 1):.3f} ms ({st['bytes'] / (ks / max(kn, 1)) / 1e12:.2f} TB/s) build {1e3 * el / steps:.3f} ms = {steps / el:.0f} builds/s | "
      f"|J| {np.abs(J).sum():.10e} |K| {np.abs(K).sum():.10e} sym {np.abs(J - J.T).max():.1e} {np.abs(K - K.T).max():.1e}", flush=True)
