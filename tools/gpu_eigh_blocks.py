"""GPU: what the symmetric eigensolvers cost at the sizes of a diatomic's symmetry blocks (N = 400: 162 / 96 / 96 / 46) against the full
matrix: rocsolver_dsyevd alone, the library's eigh() (Jacobi in LDS up to 64), and four problems in one strided-batched call."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from tuna_amd.engine import Engine
eng = Engine(0)
names = {0: "dsyevd", 3: "eigh()", 6: "dsyevdj", 8: "dsyevd x4 batched", 9: "dsyevdj x4 batched", 7: "4 GEMMs"}
for n in (46, 64, 96, 128, 162, 200, 256, 400):
    row = []
    for v in (0, 3, 6, 8, 9, 7):
        if v in (8, 9) and n > 256:
            continue
        s = C.c_double()
        rc = eng._L.tf_eigh_probe(eng._ctx, n, v, 5, C.byref(s))
        row.append(f"{names[v]} {s.value * 1e3:.3f}" if rc == 0 else f"{names[v]} failed")
    print(f"n={n}: " + "  ".join(row) + "  (ms)", flush=True)
