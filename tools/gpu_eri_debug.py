import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
from oracle import oracle as orc
eng = Engine(0)
atoms = mol.make_atoms(["N", "N"], mol.angstrom_to_bohr(1.0977)); sh = mol.build_shells(atoms, sys.argv[1] if len(sys.argv) > 1 else "cc-pVDZ"); aos = mol.expand_cartesian_aos(sh)
eng.set_basis(aos).build_eri(False)
Eg = eng.copy_eri(); Eo = orc.eri(aos)
bad = np.argwhere(np.abs(Eg - Eo) > 1e-10)
print("n bad", len(bad), "of", Eo.size, "max", np.abs(Eg - Eo).max())
sof = aos.shell_of_ao
from collections import Counter
def desc(s): return f"{'SPDFGH'[sh[s].L]}{len(sh[s].exps)}"
cnt = Counter()
for i, j, k, l in bad[:200000]:
    cnt[(desc(sof[i]), desc(sof[j]), desc(sof[k]), desc(sof[l]))] += 1
for k, v in cnt.most_common(30): print(k, v)
if len(bad):
    i, j, k, l = bad[0]; print(bad[0], Eg[i, j, k, l], Eo[i, j, k, l])
