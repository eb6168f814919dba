"""GPU: which column-part entries does a poisoned Fock build leave unwritten? (debug aid; TF_JK_POISON fills the partial buffers with
NaNs before the pass)"""
import ctypes as C, os, sys
os.environ["TF_JK_POISON"] = "5"
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from conftest import make_system
from tuna_amd.engine import Engine
from tuna_amd import spherical, _lib
import layout_model as lm
tag = sys.argv[1] if len(sys.argv) > 1 else "n2_sto3g"
atoms, shells, aos, nocc = make_system(tag)
U = spherical.transformation_matrix([s.L for s in shells])
first = np.argmax(np.abs(U) > 0, axis=1)
cls = (aos.lmn[first, 0] & 1) | ((aos.lmn[first, 1] & 1) << 1)
Lm = lm.Layout(cls, 8, 64)
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    N = eng.N
    A = np.random.default_rng(0).standard_normal((N, N)); P = A + A.T
    J, K = eng.fock_jk(P)
    L = _lib.lib()
    L.tf_debug_partials.restype = C.c_longlong; L.tf_debug_partials.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
    L.tf_debug_groups.restype = C.c_int; L.tf_debug_groups.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    g5 = np.zeros((4096, 5), dtype=np.int32)
    ng = L.tf_debug_groups(eng._ctx, g5.ctypes.data, 4096)
    g5 = g5[:ng]
    DIc = np.zeros(ng * N); n = L.tf_debug_partials(eng._ctx, 0, DIc.ctypes.data, DIc.size); DIc = DIc[:n].reshape(ng, N)
    print(tag, "N", N, "groups", ng, "classes", np.bincount(cls, minlength=4).tolist(), "cstart", Lm.cstart.tolist(), "chunks", list(zip(Lm.chunk_cls, Lm.chunk_c0, Lm.chunk_width)))
    for gi in range(ng):
        i, j0, nr, r0, c = [int(v) for v in g5[gi]]
        io = int(Lm.orig[i])
        written = ~np.isnan(DIc[gi])
        expect = np.zeros(N, dtype=bool)
        for w in range(Lm.NW):
            if Lm.task_exists(c, w, io):
                expect[Lm.chunk_c0[w]:Lm.chunk_c0[w] + Lm.chunk_width[w]] = True
        if not np.array_equal(written, expect):
            print(" group", gi, "i", i, "(orig", io, ") j0", j0, "nr", nr, "c", c, "written", np.nonzero(written)[0].tolist(), "expected", np.nonzero(expect)[0].tolist())
    # emulate the column-part reads of the reduction with the model's tables on the dumped buffer
    nrows = eng.eri_storage()["rows"]
    DJc = np.zeros(nrows * N); n = L.tf_debug_partials(eng._ctx, 2, DJc.ctypes.data, DJc.size); DJc = DJc[:n].reshape(nrows, N)
    bad = np.argwhere(np.isnan(K))
    print("GPU K NaNs", len(bad))
    Dn = np.zeros((N, N), dtype=bool)
    for gi in range(ng):
        i, j0, nr, r0, c = [int(v) for v in g5[gi]]
        io = int(Lm.orig[i])
        for y in range(N):
            wy = int(Lm.chunk_of[y])
            if Lm.task_exists(c, wy, io) and np.isnan(DIc[gi, y]):
                Dn[i, y] = True
                print("  model rule reads unwritten DIc: group", gi, "x", i, "y", y)
    print("model-rule NaN reads of DIc:", int(Dn.sum()))
    # rows
    sig = Lm.sigma
    rows = sorted(((int(sig[i]), int(sig[j])) for i in range(N) for j in range(i + 1)))
    cnt = 0
    for r, (iI, jI) in enumerate(rows):
        if iI == jI:
            continue
        c = int(Lm.clsI[iI] ^ Lm.clsI[jI]); io = int(Lm.orig[iI])
        for y in range(N):
            wy = int(Lm.chunk_of[y])
            if Lm.task_exists(c, wy, io) and np.isnan(DJc[r, y]):
                cnt += 1
                if cnt < 10:
                    print("  model rule reads unwritten DJc: row", r, (iI, jI), "y", y, "c", c)
    print("model-rule NaN reads of DJc:", cnt)
    Dd = np.zeros(N * N); L.tf_debug_partials(eng._ctx, 4, Dd.ctypes.data, Dd.size); Dd = Dd.reshape(N, N)
    print("device D NaNs (internal indices):", int(np.isnan(Dd).sum()), np.argwhere(np.isnan(Dd))[:10].tolist())
    gr = np.zeros(ng * 4); L.tf_debug_partials(eng._ctx, 5, gr.ctypes.data, gr.size); gr = gr.view(np.int32).reshape(ng, 8)
    for gi in range(min(ng, 6)):
        i = int(g5[gi, 0]); io = int(Lm.orig[i])
        print("  grec", gi, gr[gi].tolist(), "expected ke", [Lm.ke(a, io) for a in range(4)], "c", int(g5[gi, 4]))
    rr = np.zeros(nrows * 4); L.tf_debug_partials(eng._ctx, 6, rr.ctypes.data, rr.size); rr = rr.view(np.int32).reshape(nrows, 8)
    nbad = 0
    for r, (iI, jI) in enumerate(rows):
        io = int(Lm.orig[iI])
        exp = [Lm.ke(a, io) for a in range(4)] + [int(Lm.clsI[iI] ^ Lm.clsI[jI])]
        if rr[r, :5].tolist() != exp:
            nbad += 1
            if nbad < 8:
                print("  rrec mismatch row", r, (iI, jI), rr[r].tolist(), "expected", exp)
    print("rrec mismatches:", nbad, "of", nrows)
    # literal emulation of kd_reduce_block / kd_parts (column parts only) on the dumps
    DJr_dummy = None
    jptr = np.zeros(N + 1, dtype=np.int64); 
    for (iI, jI) in rows:
        if iI != jI: jptr[jI + 1] += 1
    jptr = np.cumsum(jptr)
    fill = jptr[:-1].copy(); jrows = np.zeros(max(1, jptr[-1]), dtype=np.int64)
    for r, (iI, jI) in enumerate(rows):
        if iI != jI:
            jrows[fill[jI]] = r; fill[jI] += 1
    gfirst = {}
    for gi in range(ng):
        gfirst.setdefault(int(g5[gi, 0]), []).append(gi)
    def parts(rec, y, cy, ly, wy, colrow):
        ke = rec[:4]; c = int(rec[4]); a = cy ^ c
        hit = False
        if Lm.kap0[c][wy] < ke[a]:
            hit = bool(np.isnan(colrow[y]))
        return hit
    nanreads = 0
    for x in range(N):
        for y in range(N):
            cy = int(Lm.clsI[y]); ly = y - int(Lm.cstart[cy]); wy = int(Lm.chunk_of[y])
            for gi in gfirst.get(x, []):
                if parts(gr[gi], y, cy, ly, wy, DIc[gi]): nanreads += 1; print("   literal: group read NaN", x, y, gi, gr[gi].tolist())
            for p in range(jptr[x], jptr[x + 1]):
                r = int(jrows[p])
                if parts(rr[r], y, cy, ly, wy, DJc[r]): nanreads += 1; print("   literal: row read NaN", x, y, r, rr[r].tolist()) if nanreads < 12 else None
    print("literal emulation NaN reads:", nanreads)
