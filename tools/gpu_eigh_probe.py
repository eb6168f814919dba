import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd.engine import Engine
eng = Engine(0)
for n in (60, 118, 200, 400):
    row = []
    for v in (0, 3, 4, 5):
        s = C.c_double()
        rc = eng._L.tf_eigh_probe(eng._ctx, n, v, 5, C.byref(s))
        row.append(s.value * 1e3 if rc == 0 else float("nan"))
    print(f"n={n}: dsyevd {row[0]:.3f} ms  eigh() {row[1]:.3f} ms  dsyevdx(18 lowest) {row[2]:.3f} ms  dsyevx(18 lowest) {row[3]:.3f} ms")
import numpy as np, time
for n in (60, 118, 400):
    A = np.random.default_rng(0).standard_normal((n, n)); A = A + A.T
    t = time.perf_counter()
    for _ in range(5): np.linalg.eigh(A)
    print(f"host LAPACK n={n}: {(time.perf_counter()-t)/5*1e3:.3f} ms")

# accuracy of tf_diagonalise (whatever solver it picks) against LAPACK
for n in (2, 3, 10, 28, 60, 61, 98, 118, 140, 141):
    rng = np.random.default_rng(n); A = rng.standard_normal((n, n)); F = A + A.T
    B = rng.standard_normal((n, n)); S = B @ B.T + n * np.eye(n)
    w, v = np.linalg.eigh(S); X = v @ np.diag(w ** -0.5) @ v.T
    eps, Cm = eng.diagonalise(F, X)
    Fo = X.T @ F @ X; Fo = 0.5 * (Fo + Fo.T); ref = np.linalg.eigvalsh(Fo)
    resid = np.abs(F @ Cm - S @ Cm * eps).max()
    print(f"n={n}: max|eps - lapack| = {np.abs(eps - ref).max():.2e}  residual |F C - S C eps| = {resid:.2e}  orth |C^T S C - 1| = {np.abs(Cm.T @ S @ Cm - np.eye(n)).max():.2e}")
