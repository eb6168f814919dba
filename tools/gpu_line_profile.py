"""GPU: cProfile of one warm reference-style input line (where the host time of a single point goes)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import energy
from tuna_amd.engine import Engine
line = sys.argv[1] if len(sys.argv) > 1 else "SPE : AR AR 3.76 : HF CC-PVQZ"
eng = Engine(0)
for _ in range(2):
    energy.run(line, engine=eng, silent=True)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
out = energy.run(line, engine=eng, silent=True)
pr.disable()
print("wall %.1f ms, E = %.10f, %d iterations, timings %s" % ((time.perf_counter() - t0) * 1e3, out.energy, out.n_iterations, getattr(out, "timings", None)))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
