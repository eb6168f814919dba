"""GPU: a sweep of reference-style input lines over basis sets (second run of each on a warm Engine): total and per-stage wall time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import energy
from tuna_amd.engine import Engine
LINES = ["SPE : AR AR 3.76 : HF AUG-CC-PVQZ", "SPE : AR AR 3.76 : HF CC-PV5Z", "SPE : C O 1.128 : HF DEF2-QZVP", "SPE : C O 1.128 : B3LYP DEF2-QZVP",
         "SPE : H F 0.917 : HF AUG-CC-PVQZ", "SPE : O O 1.2075 : UHF CC-PVQZ : ML 3", "SPE : AR : HF CC-PV5Z", "SPE : N N 1.0977 : MP2 CC-PVQZ",
         "SPE : CL CL 1.99 : HF CC-PVTZ", "SPE : LI H 1.6 : HF 6-311G**", "SPE : N N 1.0977 : HF CC-PVDZ : DECONTRACT"]
eng = Engine(0)
for line in (sys.argv[1:] or LINES):
    try:
        for rep in range(2):
            t0 = time.perf_counter()
            out = energy.run(line, engine=eng, silent=True)
            dt = time.perf_counter() - t0
        t = out.timings
        print("%-44s N=%4d %8.1f ms | 1e %6.1f ERI %7.1f SCF %7.1f (%2d it) | E = %.10f" % (line, eng.N, dt * 1e3, t.get("One-electron integrals", 0) * 1e3,
              t.get("Two-electron integrals", 0) * 1e3, t.get("Self-consistent field", 0) * 1e3, out.n_iterations, out.energy), flush=True)
    except Exception as e:
        print("%-44s FAILED: %s" % (line, str(e)[:150]), flush=True)
