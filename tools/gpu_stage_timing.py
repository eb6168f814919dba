import sys, time
sys.path.insert(0, "/root/repo")
import bench
from tuna_amd.engine import Engine
eng = Engine(0)
for wl in ("n2-cc-pvtz", "ar2-cc-pvqz"):
    atoms, shells, aos, nocc, desc = bench.build_workload(wl)
    eng.set_basis(aos)
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    org = [0, 0, 0.5 * atoms[-1].origin[2]]
    for _ in range(3): eng.one_electron(xyz, chg, org)
    t0 = time.perf_counter()
    for _ in range(20): S, T, V, D, Q = eng.one_electron(xyz, chg, org)
    t1 = time.perf_counter()
    for _ in range(20): X, smin, _ = eng.orthogonaliser(S)
    t2 = time.perf_counter()
    for _ in range(20): eng.set_basis(aos)
    t3 = time.perf_counter()
    print(wl, "one_electron %.3f ms, orthogonaliser %.3f ms, set_basis %.3f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t1) / 20 * 1e3, (t3 - t2) / 20 * 1e3))
