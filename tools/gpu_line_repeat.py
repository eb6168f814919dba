"""GPU: repeat one input line on a warm Engine and print the stage timings of every repetition."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import energy
from tuna_amd.engine import Engine
line = sys.argv[1] if len(sys.argv) > 1 else "SPE : AR AR 3.76 : HF CC-PVQZ"
eng = Engine(0)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    t0 = time.perf_counter()
    out = energy.run(line, engine=eng, silent=True)
    dt = time.perf_counter() - t0
    t = out.timings
    print("rep %d: %.1f ms | 1e %.1f, ERI %.1f (device: %s), SCF %.1f" % (rep, dt * 1e3, t.get("One-electron integrals", 0) * 1e3,
          t.get("Two-electron integrals", 0) * 1e3, {k: round(float(v) * 1e3, 1) for k, v in eng.eri_timings().items() if k.endswith("_s")},
          t.get("Self-consistent field", 0) * 1e3))
