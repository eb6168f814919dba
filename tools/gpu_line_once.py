"""GPU: run one input line twice on a warm Engine (for rocprofv3 traces of the second run)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import energy
from tuna_amd.engine import Engine
eng = Engine(0)
for _ in range(2):
    out = energy.run(sys.argv[1], engine=eng, silent=True)
print(out.energy)
