"""GPU: tensor build of the BASELINE basis sets (small-problem mode), team kernels against the previous path.
usage: python tools/gpu_eri_small.py [VAR=value ...]"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r"""
import json, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import bench
from tuna_amd.engine import Engine
out = {}
with Engine(0) as eng:
    for wl in ("n2-cc-pvtz", "ar2-cc-pvqz"):
        atoms, shells, aos, nocc, desc = bench.build_workload(wl)
        eng.set_basis(aos)
        best = None
        for rep in range(5):
            t0 = time.perf_counter(); eng.build_eri(True); wall = time.perf_counter() - t0
            t = eng.eri_timings()
            if rep and (best is None or wall < best["wall_ms"] * 1e-3):
                best = {"eri_kernels_ms": 1e3 * t["cart_kernel_s"], "device_total_ms": 1e3 * t["total_s"], "wall_ms": 1e3 * wall}
        idx = np.random.default_rng(0).integers(0, eng.N, size=(4000, 4)).astype(np.int32)
        best["checksum"] = float(np.abs(eng.sample_eri(idx)).sum())
        out[wl] = best
print(json.dumps(out))
"""
for name in (sys.argv[1:] or ["base"]):
    env = dict(os.environ)
    for kv in name.split(","):
        if "=" in kv:
            k, v = kv.split("=", 1); env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True)
    print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
