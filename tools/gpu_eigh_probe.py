import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd.engine import Engine
eng = Engine(0)
for n in (30, 60, 118, 200, 400):
    row = []
    for v in (0, 1, 2):
        s = C.c_double()
        rc = eng._L.tf_eigh_probe(eng._ctx, n, v, 5, C.byref(s))
        row.append(s.value * 1e3 if rc == 0 else float("nan"))
    print(f"n={n}: dsyevd {row[0]:.3f} ms  dsyev {row[1]:.3f} ms  dsyevj {row[2]:.3f} ms")
import numpy as np, time
for n in (60, 118, 400):
    A = np.random.default_rng(0).standard_normal((n, n)); A = A + A.T
    t = time.perf_counter()
    for _ in range(5): np.linalg.eigh(A)
    print(f"host LAPACK n={n}: {(time.perf_counter()-t)/5*1e3:.3f} ms")
