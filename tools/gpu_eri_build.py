"""GPU: build the tensor of a bench workload a few times (the process rocprofv3 wraps for ERI profiles).
usage: python tools/gpu_eri_build.py [workload=synth-400] [builds=2]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from tuna_amd.engine import Engine
wl = sys.argv[1] if len(sys.argv) > 1 else "synth-400"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
atoms, shells, aos, nocc, desc = bench.build_workload(wl)
with Engine(0) as eng:
    eng.set_basis(aos)
    for rep in range(n):
        t0 = time.perf_counter(); eng.build_eri(True); wall = time.perf_counter() - t0
        print(json.dumps(dict(eng.eri_timings(), wall_s=wall, counts=eng.eri_counts() if hasattr(eng, "eri_counts") else None)), flush=True)
