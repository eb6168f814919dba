"""TEST INFRASTRUCTURE: a NumPy model of the "tiles" tensor layout and of the Fock-build algebra that
tuna_amd/csrc/tf_jktile.hip.h runs on it.  The tables (regions, tasks, shapes of the partial sums) are NOT restated here: they come
from the library's own host builder (tuna_amd/csrc/tf_tiles_host.h), compiled for the CPU by tests/tile_model/build.sh, so that the
index algebra the GPU kernels rely on is checked on a CPU against the reference einsums (scf:55-72 "ijkl,kl->ij", scf:27-44
"ilkj,kl->ij").  Nothing in the product imports this module.

The model executes every task step by step the way a workgroup does (waves = column blocks, rows = the strip's k, one j per step),
writes the same partial-sum arrays at the same offsets and then runs the same reductions.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class TTask(C.Structure):
    _fields_ = [("base", C.c_longlong), ("jt_base", C.c_longlong), ("dj_base", C.c_longlong), ("i", C.c_int), ("j0", C.c_int), ("nj", C.c_int),
                ("a", C.c_int), ("b", C.c_int), ("k0", C.c_int), ("nks", C.c_int), ("roff0", C.c_int), ("lb0", C.c_int), ("nw", C.c_int),
                ("nk", C.c_int), ("nl", C.c_int), ("slice", C.c_int), ("woff", C.c_int * 4), ("jt_pitch", C.c_int), ("dj_len", C.c_int),
                ("dj_koff", C.c_int), ("dj_loff", C.c_int * 4), ("di_base", C.c_int), ("jd_base", C.c_int), ("self_last", C.c_int), ("pid", C.c_int),
                ("kbase", C.c_int), ("lbase", C.c_int), ("ncol", C.c_int), ("pm_off", C.c_int), ("pm_pitch", C.c_int), ("pad_", C.c_int)]


class TPairI(C.Structure):
    _fields_ = [("first_task", C.c_int), ("nparts", C.c_int), ("tasks_per_part", C.c_int), ("j0", C.c_int), ("nj", C.c_int), ("pj", C.c_int),
                ("nk", C.c_int), ("nl", C.c_int), ("jt_base", C.c_longlong), ("jt_part_stride", C.c_longlong), ("jt_pitch", C.c_int),
                ("dj_k", C.c_int), ("dj_l", C.c_int)]


class TRunI(C.Structure):
    _fields_ = [("j0", C.c_int), ("nj", C.c_int), ("dj_len", C.c_int), ("pad", C.c_int), ("dj_base", C.c_longlong), ("e_base", C.c_longlong)]


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "tile_model", "_build", "libtiletables.so")
        src = [os.path.join(HERE, "tile_model", "tile_tables.cpp"), os.path.join(HERE, "..", "tuna_amd", "csrc", "tf_tiles.h"),
               os.path.join(HERE, "..", "tuna_amd", "csrc", "tf_tiles_host.h")]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
            subprocess.check_call(["sh", os.path.join(HERE, "tile_model", "build.sh")])
        L = C.CDLL(so)
        L.ttm_build.restype = C.c_void_p
        L.ttm_build.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.ttm_free.argtypes = [C.c_void_p]
        L.ttm_error.restype = C.c_char_p
        L.ttm_error.argtypes = [C.c_void_p]
        L.ttm_count.restype = C.c_longlong
        L.ttm_count.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.ttm_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ttm_elem_addr.restype = C.c_longlong
        L.ttm_elem_addr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        _LIB = L
    return _LIB


class Tables:
    """the host tables of one build: cls[N] (original AO order), the owned rows (original (i >= j)), the strip height of the task list"""

    def __init__(self, cls, rows, ksub=64, part_steps=48):
        L = lib()
        cls = np.ascontiguousarray(cls, dtype=np.int32)
        rows = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1, 2)
        self.N = N = len(cls)
        self.cls = cls
        self.ksub = ksub
        h = L.ttm_build(N, cls.ctypes.data, len(rows), rows.ctypes.data, ksub, part_steps)
        self._h = h
        try:
            err = L.ttm_error(h).decode()
            if err:
                raise ValueError(err)
            which = 0 if ksub == 64 else 1
            cnt = lambda what, w=which: int(L.ttm_count(h, w, what))
            assert cnt(9) == C.sizeof(TTask) and cnt(10) == C.sizeof(TPairI) and cnt(11) == C.sizeof(TRunI)

            def arr(ctype, n, what, w=which):
                buf = (ctype * max(n, 1))()
                L.ttm_copy(h, w, what, buf)
                return buf

            def ints(n, what, w=which):
                a = np.zeros(max(n, 1), dtype=np.int32)
                L.ttm_copy(h, w, what, a.ctypes.data)
                return a[:n]

            self.tasks = arr(TTask, cnt(0), 0)
            self.n_tasks = cnt(0)
            self.pairs = arr(TPairI, N * 10, 1)
            self.runs = arr(TRunI, N * 4, 2)
            self.regions = arr(TTask, cnt(8), 3)
            self.n_regions = cnt(8)
            self.prim_pairs = arr(TPairI, N * 10, 1, 0)
            self.n_elems, self.edge_base = cnt(1), cnt(2)
            self.dj_len, self.jd_len, self.jt_len, self.n_di, self.npair = cnt(3), cnt(4), cnt(5), cnt(6), cnt(7)
            self.sigma, self.origI, self.clsI = ints(N, 4), ints(N, 5), ints(N, 6)
            self.cstart = ints(4, 7)
            self.pa, self.pb = ints(10, 8), ints(10, 9)
            self.itask_ptr = ints(N + 1, 10)
            self.itasks = ints(self.n_tasks, 11)
            self.jlist_ptr = ints(N + 1, 12)
            self.jlist = ints(cnt(12), 13)
            self.bucket = ints(5, 14)
            self.cntA = ints(4 * N, 15).reshape(4, N)
            self.csize = np.bincount(cls, minlength=4)
        except Exception:
            L.ttm_free(h)
            self._h = None
            raise

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ttm_free(self._h)
            self._h = None

    def elem_addr(self, iI, jI, kI, lI):
        """tt_elem_addr of tf_tiles.h (the address function of the writer and of the element accessors)"""
        return int(lib().ttm_elem_addr(self._h, int(iI), int(jI), int(kI), int(lI)))

    # shape functions of tf_tiles.h
    def row_len(self, tri, ks, lb, r, nl): return lib().ttm_row_len(int(tri), ks, lb, r, nl)
    def row_off(self, tri, ks, lb, r, nl): return lib().ttm_row_off(int(tri), ks, lb, r, nl)
    def elem_off(self, tri, ks, lb, r, c, nks, nl): return lib().ttm_elem_off(int(tri), ks, lb, r, c, nks, nl)
    def dj_koff(self, tri, ks, nl): return lib().ttm_dj_koff(int(tri), ks, nl)
    def dj_loff(self, tri, lb, nk): return lib().ttm_dj_loff(int(tri), lb, nk, self.ksub)
    def dj_first_sub(self, tri, lb): return lib().ttm_dj_first_sub(int(tri), lb, self.ksub)
    def nlb(self, tri, ks, nk, nl): return lib().ttm_nlb(int(tri), ks, nk, nl)

    def chunks(self, nlb):
        a, b = C.c_int(), C.c_int()
        lib().ttm_chunks(nlb, C.byref(a), C.byref(b))
        return a.value, b.value

    def loc(self, xI): return int(xI - self.cstart[self.clsI[xI]])

    def below(self, a, iI): return int(self.cntA[a, iI] - (1 if a == self.clsI[iI] else 0))


def pack_tensor(T: Tables, E):
    """the stored tensor: interior regions [j][wave piece][row][l] and the edge elements E[i][j][l]; pads are zero"""
    buf = np.zeros(T.n_elems)
    written = np.zeros(T.n_elems, dtype=bool)
    o = T.origI
    for q in range(T.n_regions):
        t = T.regions[q]
        tri = t.a == t.b
        ks = t.k0 // 64
        io = o[t.i]
        for s in range(t.nj):
            jo = o[t.j0 + s]
            for w in range(t.nw):
                lb = t.lb0 + w
                for r in range(t.nks):
                    ko = o[T.cstart[t.a] + t.k0 + r]
                    n = T.row_len(tri, ks, lb, r, t.nl)
                    for u in range(n):
                        ll = 16 * lb + u
                        off = t.base + s * t.slice + t.woff[w] + T.elem_off(tri, ks, lb, r, u, t.nks, t.nl)
                        assert t.woff[w] <= off - t.base - s * t.slice < (t.woff[w + 1] if w + 1 < t.nw else t.slice)
                        assert not written[off]
                        written[off] = True
                        if ll < (t.k0 + r + 1 if tri else t.nl):
                            buf[off] = E[io, jo, ko, o[T.cstart[t.b] + ll]]
    for iI in range(T.N):
        for cj in range(4):
            R = T.runs[iI * 4 + cj]
            for s in range(R.nj):
                jI = R.j0 + s
                lj, l0 = T.loc(jI), T.loc(R.j0)
                for ll in range(lj + 1):
                    off = T.edge_base + R.e_base + lj * (lj + 1) // 2 - l0 * (l0 + 1) // 2 + ll
                    assert not written[off]
                    written[off] = True
                    buf[off] = E[o[iI], o[jI], o[iI], o[T.cstart[cj] + ll]]
    return buf, written


def stored_count(T: Tables):
    """unique parity-allowed elements the owned rows must hold (canonical form)"""
    n = 0
    for iI in range(T.N):
        for cj in range(4):
            R = T.runs[iI * 4 + cj]
            if not R.nj:
                continue
            c = T.clsI[iI] ^ cj
            for p in range(T.npair):
                a, b = T.pa[p], T.pb[p]
                if a ^ b != c:
                    continue
                nk, nl = T.below(a, iI), T.below(b, iI)
                n += R.nj * (nk * (nk + 1) // 2 if a == b else nk * nl)
            l0 = T.loc(R.j0)
            n += (l0 + R.nj) * (l0 + R.nj + 1) // 2 - l0 * (l0 + 1) // 2
    return n


def fock(T: Tables, buf, P):
    """J and K (original indices) of the owned rows' part of the tensor, the way the kernels compute them"""
    N, o = T.N, T.origI
    X = P[np.ix_(o, o)]                                     # internal indices
    assert np.array_equal(X, X.T)
    Ppair = X + X.T - np.diag(np.diag(X))
    DJv = np.full(max(T.dj_len, 1), np.nan)
    Jtp = np.full(max(T.jt_len, 1), np.nan)
    Jdp = np.full(max(T.jd_len, 1), np.nan)
    DIk = np.full((max(T.n_di, 1), 64), np.nan)
    DIl = np.full((max(T.n_di, 1), 16), np.nan)
    for q in range(T.n_tasks):
        t = T.tasks[q]
        tri = t.a == t.b
        ks = t.k0 // 64
        kI = T.cstart[t.a] + t.k0 + np.arange(t.nks)
        jt = np.zeros((t.nw, t.nks, 16)); dik = np.zeros((t.nw, t.nks)); dil = np.zeros((t.nw, 16))
        for s in range(t.nj):
            jI = t.j0 + s
            self_step = bool(t.self_last) and s == t.nj - 1
            assert self_step == (jI == t.i)
            r1 = np.zeros(t.nks)
            for w in range(t.nw):
                lb = t.lb0 + w
                lI = T.cstart[t.b] + 16 * lb + np.arange(16)
                lval = (16 * lb + np.arange(16)) < T.csize[t.b]
                lIc = np.where(lval, lI, 0)
                m = np.zeros((t.nks, 16))
                nst = min(64, t.nk - 64 * ks)
                for r in range(t.nks):
                    n = T.row_len(tri, ks, lb, t.roff0 + r, t.nl)
                    for u in range(n):
                        m[r, u] = buf[t.base + s * t.slice + t.woff[w] + T.elem_off(tri, ks, lb, t.roff0 + r, u, nst, t.nl)]
                assert not m[:, ~lval].any()
                mo = np.where(kI[:, None] == lI[None, :], 0.0, m)         # without the diagonal k == l
                Jdp[t.jd_base + w * t.nj + s] = (m * Ppair[np.ix_(kI, lIc)]).sum()
                jt[w] += m * Ppair[t.i, jI]
                dik[w] += m @ X[jI, lIc]                                   # D[i][k] += m P[j][l]
                dil[w] += mo.T @ X[jI, kI]                                 # D[i][l] += m P[j][k]   (k != l)
                r1 += m @ X[t.i, lIc]                                      # D[j][k] += m P[i][l]   (i != j)
                r4 = mo.T @ X[t.i, kI]                                     # D[j][l] += m P[i][k]   (i != j, k != l)
                if not self_step:
                    DJv[t.dj_base + s * t.dj_len + t.dj_loff[w] + np.arange(16)] = r4
            if not self_step:
                DJv[t.dj_base + s * t.dj_len + t.dj_koff + np.arange(t.nks)] = r1
        for w in range(t.nw):
            lb = t.lb0 + w
            for r in range(t.nks):
                for u in range(16):
                    if 16 * lb + u < t.jt_pitch:
                        Jtp[t.jt_base + (t.k0 + r) * t.jt_pitch + 16 * lb + u] = jt[w, r, u]
            DIk[t.di_base + w, :t.nks] = dik[w]
            DIl[t.di_base + w] = dil[w]
    # ---- reductions
    D = np.zeros((N, N)); Jint = np.zeros((N, N))           # D[x][y] internal; Jint[hi][lo] for the pair (hi >= lo in ORIGINAL order)
    # Jt: per pair (k, l) the sum over the first indices above it, parts in order
    for iI in range(N):
        for p in range(T.npair):
            Pr = T.pairs[iI * 10 + p]
            if Pr.first_task < 0:
                continue
            a, b = T.pa[p], T.pb[p]
            for part in range(Pr.nparts):
                blk = Jtp[Pr.jt_base + part * Pr.jt_part_stride: Pr.jt_base + (part + 1) * Pr.jt_part_stride].reshape(Pr.nk, Pr.jt_pitch)
                for kl in range(Pr.nk):
                    for ll in range(kl + 1 if a == b else Pr.nl):
                        kI_, lI_ = T.cstart[a] + kl, T.cstart[b] + ll
                        hi, lo = (kI_, lI_) if o[kI_] >= o[lI_] else (lI_, kI_)
                        Jint[hi, lo] += blk[kl, ll]
    # DJ: D[j][x] over the rows (i, j), i != j
    for jI in range(N):
        cj = T.clsI[jI]
        for q in range(T.jlist_ptr[jI], T.jlist_ptr[jI + 1]):
            iI = T.jlist[q]
            R = T.runs[iI * 4 + cj]
            vec = R.dj_base + (jI - R.j0) * R.dj_len
            c = T.clsI[iI] ^ cj
            for p in range(T.npair):
                a, b = T.pa[p], T.pb[p]
                if a ^ b != c:
                    continue
                Pr = T.pairs[iI * 10 + p]
                if Pr.first_task < 0:
                    continue
                tri = a == b
                for kl in range(Pr.nk):                              # x as a row index k
                    ks = kl // 64
                    nst = min(64, Pr.nk - 64 * ks)
                    nch, _ = T.chunks(T.nlb(tri, ks, Pr.nk, Pr.nl))
                    for ch in range(nch):
                        D[jI, T.cstart[a] + kl] += DJv[vec + Pr.dj_k + T.dj_koff(tri, ks, Pr.nl) + ch * nst + kl % 64]
                ns = (Pr.nk + T.ksub - 1) // T.ksub
                for ll in range(Pr.nl):                              # x as a column index l
                    lb = ll // 16
                    for sub in range(T.dj_first_sub(tri, lb), ns):
                        D[jI, T.cstart[b] + ll] += DJv[vec + Pr.dj_l + T.dj_loff(tri, lb, Pr.nk) + (sub - T.dj_first_sub(tri, lb)) * 16 + ll % 16]
            # the edge elements' D[j][l] += w (ij|il) P[i][i]
            lj, l0 = T.loc(jI), T.loc(R.j0)
            eb = T.edge_base + R.e_base + lj * (lj + 1) // 2 - l0 * (l0 + 1) // 2
            for ll in range(lj + 1):
                D[jI, T.cstart[cj] + ll] += (0.5 if ll == lj else 1.0) * buf[eb + ll] * X[iI, iI]
    # gather: D[i][x] and Jd[i][j] of the tasks of i
    for iI in range(N):
        for q in range(T.itask_ptr[iI], T.itask_ptr[iI + 1]):
            t = T.tasks[T.itasks[q]]
            assert t.i == iI
            for w in range(t.nw):
                D[iI, T.cstart[t.a] + t.k0: T.cstart[t.a] + t.k0 + t.nks] += DIk[t.di_base + w, :t.nks]
                for u in range(16):
                    if 16 * (t.lb0 + w) + u < T.csize[t.b]:
                        D[iI, T.cstart[t.b] + 16 * (t.lb0 + w) + u] += DIl[t.di_base + w, u]
                for s in range(t.nj):
                    Jint[iI, t.j0 + s] += Jdp[t.jd_base + w * t.nj + s]
    # edge: m = (ij|il), l <= j
    for iI in range(N):
        for cj in range(4):
            R = T.runs[iI * 4 + cj]
            l0 = T.loc(R.j0) if R.nj else 0
            for s in range(R.nj):
                jI = R.j0 + s
                lj = T.loc(jI)
                eb = T.edge_base + R.e_base + lj * (lj + 1) // 2 - l0 * (l0 + 1) // 2
                for ll in range(lj + 1):
                    lI = T.cstart[cj] + ll
                    m = buf[eb + ll]
                    wgt = 0.5 if lI == jI else 1.0
                    Jint[iI, jI] += m * Ppair[iI, lI]
                    if lI != jI:
                        Jint[iI, lI] += m * Ppair[iI, jI]
                    D[iI, iI] += wgt * m * X[jI, lI]
                    if lI != iI:
                        D[iI, lI] += wgt * m * X[jI, iI]
                    if jI != iI:
                        D[jI, iI] += wgt * m * X[iI, lI]
    assert not np.isnan(D).any() and not np.isnan(Jint).any()
    Kint = D + D.T
    s = T.sigma
    K = Kint[np.ix_(s, s)]
    Jo = np.zeros((N, N))
    for x in range(N):
        for y in range(x + 1):
            Jo[x, y] = Jo[y, x] = Jint[s[x], s[y]]
    return Jo, K


def random_parity_tensor(cls, seed=0):
    """dense [N,N,N,N] with the 8-fold symmetry and the parity zeros of a z-axis diatomic"""
    cls = np.asarray(cls)
    N = len(cls)
    rng = np.random.default_rng(seed)
    ii, jj = np.tril_indices(N)
    npair = len(ii)
    A = rng.standard_normal((npair, npair))
    A = A + A.T
    pc = cls[ii] ^ cls[jj]
    A[pc[:, None] != pc[None, :]] = 0.0
    pidx = np.zeros((N, N), dtype=np.int64)
    pidx[ii, jj] = np.arange(npair); pidx[jj, ii] = np.arange(npair)
    return A[pidx[:, :, None, None], pidx[None, None, :, :]]
