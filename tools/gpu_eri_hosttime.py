"""GPU: host-side timeline of tf_build_eri (TF_DEBUG stamps) on a warm context, for one workload of bench.py."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from tuna_amd.engine import Engine
atoms, shells, aos, nocc, desc = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "ar2-cc-pvqz")
eng = Engine(0)
eng.set_basis(aos)
for rep in range(3):
    t0 = time.perf_counter()
    eng.build_eri(True)
    print("build_eri wall %.2f ms" % ((time.perf_counter() - t0) * 1e3), eng.eri_timings(), file=sys.stderr)
