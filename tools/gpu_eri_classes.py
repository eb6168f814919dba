"""GPU: time of every ERI class launch of the bench workload run alone (TF_ERI_CLASS_TIMES=1), summed per class.
usage: python tools/gpu_eri_classes.py [N]"""
import os, re, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
n = int(sys.argv[2])
atoms = mol.make_atoms(["AR", "AR"], 7.1)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*mol.synthetic_counts(n))}))
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
"""
n = sys.argv[1] if len(sys.argv) > 1 else "400"
out = subprocess.run([sys.executable, "-c", CHILD, ROOT, n], env=dict(os.environ, TF_ERI_CLASS_TIMES="1"), capture_output=True, text=True)
acc = {}
for m in re.finditer(r"\[tf eri class\] \((\d) (\d)\|(\d) (\d)\) npq (\d+) bra (\d+) ket (\d+) quartets (\d+): ([\d.]+) ms", out.stderr):
    key = tuple(int(x) for x in m.groups()[:4])
    q, ms = float(m.group(8)), float(m.group(9))
    a = acc.setdefault(key, [0.0, 0.0, 0])
    a[0] += ms; a[1] += q; a[2] += 1
tot = sum(a[0] for a in acc.values())
print("total %.1f ms over %d classes (launches run one at a time)" % (tot, len(acc)))
for key, (ms, q, nl) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:40]:
    print("(%d %d|%d %d)  %7.2f ms  %5.1f %%  %9.0f quartets  %7.1f ns/quartet  %d launches" % (*key, ms, 100 * ms / tot, q, 1e6 * ms / q, nl))
if not acc:
    print(out.stderr[-2000:])
