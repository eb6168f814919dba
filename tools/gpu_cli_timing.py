"""GPU: wall time of the reference-style input lines of the five BASELINE configurations on one warm Engine (second run of each line)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import energy
from tuna_amd.engine import Engine

LINES = ["SPE : H H 0.74 : HF STO-3G", "SPE : N N 1.0977 : HF CC-PVTZ", "SPE : AR AR 3.76 : HF CC-PVQZ", "SPE : C O 1.128 : B3LYP DEF2-TZVP",
         "SPE : N N 1.0977 : MP2 CC-PVTZ", "SPE : O O 1.2075 : UHF CC-PVDZ : ML 3", "SPE : O O 1.2075 : UHF CC-PVTZ : ML 3",
         "SPE : N O 1.151 : UHF CC-PVQZ : ML 2"]
eng = Engine(0)
for line in LINES:
    for rep in range(2):
        t0 = time.perf_counter()
        res = energy.run(line, engine=eng, silent=True) if "engine" in energy.run.__code__.co_varnames else energy.run(line)
        dt = time.perf_counter() - t0
    e = getattr(res, "energy", None) if not isinstance(res, dict) else res.get("energy")
    print(f"{line:42s} warm {dt*1e3:8.1f} ms   E = {e}")

# the unrestricted cycle: whole cycle in the library against the host-orchestrated loop
for line in LINES[5:]:
    os.environ["TUNA_AMD_HOST_UHF"] = "1"
    for rep in range(2):
        t0 = time.perf_counter()
        res = energy.run(line, engine=eng, silent=True) if "engine" in energy.run.__code__.co_varnames else energy.run(line)
        dt = time.perf_counter() - t0
    del os.environ["TUNA_AMD_HOST_UHF"]
    print(f"{line:42s} host-orchestrated warm {dt*1e3:8.1f} ms   E = {res.energy}  ({res.n_iterations} iterations)")
