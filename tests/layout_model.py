"""TEST INFRASTRUCTURE: a NumPy model of the parity-blocked packed tensor layout and of the Fock-build algebra that
tuna_amd/csrc/tf_jkpacked.hip.h runs on it (tables, tasks, partial sums, validity rules of the reductions).  It mirrors the
host-side table construction of tf_build_eri (tuna_amd/csrc/tf_device.hip: build_blocked_layout / build_jk_tables) so that the
index algebra can be checked on a CPU against the reference einsums (scf:55-72, scf:27-44); nothing in the product imports it.

Layout in one paragraph.  On a z-axis diatomic every AO has a definite parity under x -> -x and y -> -y: class 0..3 =
(x parity) | (y parity) << 1.  (ij|kl) vanishes unless class(i) ^ class(j) == class(k) ^ class(l) (the rule the reference uses to
skip work, pyx:1324-1327).  Tensor row (i >= j), of class c = class(i) ^ class(j), therefore stores for every k <= i ONE segment:
the values (ij|kl) for the AOs l <= k of class class(k) ^ c, in ascending l, padded to PAD doubles (in the segment of k == i the
slots beyond l == j hold zeros).  The segments of a row are ordered by (class of k, k): four sections, so that a task -- one
class of k against one class of l -- streams contiguous memory.  Unpadded that is the 8-fold unique part of the non-zero quarter
of the tensor.
"""
from __future__ import annotations

import numpy as np

PAD = 8          # segment alignment in doubles (TF_SEG_PAD)
CW = 64          # columns per chunk (TF_JKP_CW)
JBB = 8          # rows per group (one density per pass)
W = 8            # groups per workgroup (two per wave)


def pad_up(n, pad=PAD):
    return (n + pad - 1) // pad * pad


class Layout:
    """Tables of the blocked layout for AO classes cls[N] (original AO order)."""

    def __init__(self, cls, pad=PAD, cw=CW):
        cls = np.asarray(cls, dtype=np.int64)
        N = len(cls)
        self.N, self.cls, self.pad, self.cw = N, cls, pad, cw
        size = np.bincount(cls, minlength=4)
        order = sorted(range(4), key=lambda c: (-size[c], c))            # classes by descending size (ties: class id)
        self.cstart = np.zeros(4, dtype=np.int64)
        s = 0
        for c in order:
            self.cstart[c] = s
            s += size[c]
        self.csize = size
        self.loc = np.zeros(N, dtype=np.int64)                            # rank of an AO among the AOs of its class (original order)
        seen = [0, 0, 0, 0]
        for k in range(N):
            self.loc[k] = seen[cls[k]]
            seen[cls[k]] += 1
        self.sigma = self.cstart[cls] + self.loc                          # original -> internal (class-sorted) index
        self.orig = np.zeros(N, dtype=np.int64)
        self.orig[self.sigma] = np.arange(N)
        self.clsI = cls[self.orig]                                        # class by internal index
        # cnt[b][k]: class-b AOs with original index <= k
        self.cnt = np.zeros((4, N), dtype=np.int64)
        for b in range(4):
            self.cnt[b] = np.cumsum(cls == b)
        # per row class c and AO k (tables by internal index): segment length cntI, offset offA of the segment inside the section
        # of k's class (prefix over the members of that class), and the complete sections' starts fullsec (the "full row" shape that
        # the packed density and the Jt partials use: pair index = cbase[c] + fullsec[c][a] + offA[c][k'] + loc[l])
        self.cntI = np.zeros((4, N), dtype=np.int64)
        self.offA = np.zeros((4, N), dtype=np.int64)
        self.offE = np.zeros((4, N), dtype=np.int64)                      # offA + padded segment length
        self.fullsec = np.zeros((4, 4), dtype=np.int64)
        self.NP = np.zeros(4, dtype=np.int64)                             # padded pair-index space of each class
        self.corder = order
        for c in range(4):
            tot = 0
            for a in order:
                self.fullsec[c][a] = tot
                off = 0
                for kk in range(size[a]):
                    kI = self.cstart[a] + kk
                    n = self.cnt[a ^ c][self.orig[kI]]
                    self.cntI[c][kI] = n
                    self.offA[c][kI] = off
                    off += pad_up(n, pad)
                    self.offE[c][kI] = off
                tot += off
            self.NP[c] = tot
        self.cbase = np.concatenate([[0], np.cumsum(self.NP)[:-1]])
        self.NPtot = int(self.NP.sum())
        # granule table of the pair index space: AO k' of the segment that holds granule g (PAD doubles) of class c
        self.gk = [np.zeros(self.NP[c] // pad, dtype=np.int64) for c in range(4)]
        for c in range(4):
            for kI in range(N):
                a = self.clsI[kI]
                g0 = (self.fullsec[c][a] + self.offA[c][kI]) // pad
                g1 = (self.fullsec[c][a] + self.offE[c][kI]) // pad
                self.gk[c][g0:g1] = kI
        # chunks: the internal columns cut at class boundaries and every cw columns
        self.chunk_cls, self.chunk_lam0, self.chunk_c0, self.chunk_width = [], [], [], []
        self.wfirst = np.zeros(5, dtype=np.int64)                         # chunks of class b: wfirst[b] .. wfirst[b + 1] (class id order)
        for b in range(4):
            self.wfirst[b] = len(self.chunk_cls)
            for lam0 in range(0, size[b], cw):
                self.chunk_cls.append(b); self.chunk_lam0.append(lam0)
                self.chunk_c0.append(self.cstart[b] + lam0); self.chunk_width.append(min(cw, size[b] - lam0))
        self.wfirst[4] = len(self.chunk_cls)
        self.NW = len(self.chunk_cls)
        self.chunk_of = np.zeros(N, dtype=np.int64)                       # chunk of an internal column
        for w in range(self.NW):
            self.chunk_of[self.chunk_c0[w]: self.chunk_c0[w] + self.chunk_width[w]] = w
        # kap0[c][w]: first member (by rank in its class) of the k class a = b(w) ^ c whose segment reaches the chunk
        self.kap0 = np.zeros((4, self.NW), dtype=np.int64)
        self.rpoff = np.zeros((4, self.NW), dtype=np.int64)               # offset of the row parts of chunk w in a row's part vector
        self.RS = 0
        for c in range(4):
            o = 0
            for w in range(self.NW):
                a = self.chunk_cls[w] ^ c
                kap = size[a]
                for kk in range(size[a]):
                    if self.cntI[c][self.cstart[a] + kk] > self.chunk_lam0[w]:
                        kap = kk
                        break
                self.kap0[c][w] = kap
                self.rpoff[c][w] = o
                o += size[a]
            self.RS = max(self.RS, o)

    def seclen(self, c, a, i):
        """doubles of section a in a class-c row with first index i (original): the segments of the class-a AOs k <= i"""
        ke = self.ke(a, i)
        return 0 if ke == 0 else int(self.offE[c][self.cstart[a] + ke - 1])

    def secoff(self, c, i):
        out, tot = [0, 0, 0, 0], 0
        for a in self.corder:
            out[a] = tot
            tot += self.seclen(c, a, i)
        return out, tot

    def row_len(self, i, j):
        """stored doubles of row (i >= j), original indices: the same for every j of one class"""
        return self.secoff(int(self.cls[i] ^ self.cls[j]), i)[1]

    def pair_index(self, c, kI, lam):
        return int(self.fullsec[c][self.clsI[kI]] + self.offA[c][kI] + lam)

    def ke(self, a, i):
        """members of class a with original index <= i"""
        return int(self.cnt[a][i])

    def task_exists(self, c, w, i):
        return self.kap0[c][w] < self.ke(self.chunk_cls[w] ^ c, i)

    def key(self, x, y):
        hi, lo = max(x, y), min(x, y)
        return hi * (hi + 1) // 2 + lo


def fock_partial(L: Layout, E, P, owned_rows):
    """J, K contributions of the rows `owned_rows` (list of original (i, j), i >= j) through the kernel's task structure."""
    N = L.N
    cls, loc, sigma, orig = L.cls, L.loc, L.sigma, L.orig
    assert np.allclose(P, P.T)
    X = P[np.ix_(orig, orig)]                                             # internal-order density
    # rows sorted by internal (i', j')
    rows = sorted(((int(sigma[i]), int(sigma[j])) for (i, j) in owned_rows))
    rowmap = {L.key(a, b): r for r, (a, b) in enumerate(rows)}
    # groups: runs of consecutive j' with the same i' and the same class of j, longest rows first (built from the end)
    groups = []
    r = len(rows) - 1
    while r >= 0:
        r0 = r
        while (r0 > 0 and rows[r0 - 1][0] == rows[r][0] and rows[r0 - 1][1] == rows[r0][1] - 1
               and L.clsI[rows[r0 - 1][1]] == L.clsI[rows[r][1]] and r - r0 + 1 < JBB):
            r0 -= 1
        ii, jj0 = rows[r0]
        groups.append(dict(i=ii, j0=jj0, nr=r - r0 + 1, r0=r0, c=int(L.clsI[ii] ^ L.clsI[jj0])))
        r = r0 - 1
    gfirst = {}
    for gi, g in enumerate(groups):
        gfirst.setdefault(g["i"], []).append(gi)
    supers = []
    gi = 0
    while gi < len(groups):
        ge = gi + 1
        while ge < len(groups) and groups[ge]["i"] == groups[gi]["i"] and groups[ge]["c"] == groups[gi]["c"] and ge - gi < W:
            ge += 1
        g = groups[gi]
        io, jo = orig[g["i"]], orig[g["j0"] + g["nr"] - 1]
        supers.append(dict(g0=gi, ng=ge - gi, c=g["c"], io=int(io)))
        gi = ge
    nan = np.nan
    Jd = np.full((len(rows), L.NW), nan)
    DIc = np.full((len(groups), N), nan); DIr = np.full((len(groups), L.RS), nan)
    DJc = np.full((len(rows), N), nan); DJr = np.full((len(rows), L.RS), nan)
    ypart = [np.full(L.NP[s["c"]], nan) for s in supers]            # full-row shape: slot = pair index

    def Ppair(k, l):                                                      # original indices
        return P[k, k] if k == l else P[k, l] + P[l, k]

    n_tasks = 0
    for si, sg in enumerate(supers):
        c = sg["c"]
        i_int = groups[sg["g0"]]["i"]
        i = int(orig[i_int])
        for w in range(L.NW):
            if not L.task_exists(c, w, i):
                continue
            n_tasks += 1
            b = L.chunk_cls[w]; a = b ^ c
            lam0, width, c0 = L.chunk_lam0[w], L.chunk_width[w], L.chunk_c0[w]
            KE = L.ke(a, i)
            glist = [groups[gi] for gi in range(sg["g0"], sg["g0"] + sg["ng"])]
            colI = [np.zeros(width) for _ in glist]
            colJ = [np.zeros((g["nr"], width)) for g in glist]
            jd = [np.zeros(g["nr"]) for g in glist]
            for kap in range(L.kap0[c][w], KE):
                kI = int(L.cstart[a] + kap); k = int(orig[kI])
                cntk = int(L.cntI[c][kI])
                rI = [0.0 for _ in glist]
                rJ = [np.zeros(g["nr"]) for g in glist]
                for u in range(width):
                    lam = lam0 + u
                    if lam >= cntk:
                        break
                    lI = c0 + u; l = int(orig[lI])
                    assert l <= k and cls[l] == b and loc[l] == lam
                    jt = 0.0
                    for t, g in enumerate(glist):
                        for rr in range(g["nr"]):
                            jI = g["j0"] + rr; j = int(orig[jI])
                            if k == i and l > j:
                                continue
                            m = E[i, j, k, l]
                            same = (k == i and l == j)
                            mk = 0.5 * m if same else m
                            jd[t][rr] += m * Ppair(k, l)
                            if not same:
                                jt += m * Ppair(i, j)
                            rI[t] += mk * X[jI, lI]
                            rJ[t][rr] += mk * X[i_int, lI]
                            if l != k:
                                colI[t][u] += mk * X[jI, kI]
                                colJ[t][rr, u] += mk * X[i_int, kI]
                    ypart[si][L.pair_index(c, kI, lam)] = jt                    # one writer per slot
                for t, g in enumerate(glist):                                 # every walked k writes its row parts (zeros included)
                    DIr[sg["g0"] + t, L.rpoff[c][w] + kap] = rI[t]
                    for rr in range(g["nr"]):
                        DJr[g["r0"] + rr, L.rpoff[c][w] + kap] = rJ[t][rr]
            for t, g in enumerate(glist):
                DIc[sg["g0"] + t, c0:c0 + width] = colI[t]
                for rr in range(g["nr"]):
                    DJc[g["r0"] + rr, c0:c0 + width] = colJ[t][rr]
                    Jd[g["r0"] + rr, w] = jd[t][rr]

    # ---- reductions (validity rules of jk_reduce_kernel / jk_packed_final_kernel) ----
    Jt = np.zeros(L.NPtot)
    for c in range(4):
        for q in range(L.NP[c]):
            kI = int(L.gk[c][q // L.pad]); a = int(L.clsI[kI]); kap = kI - int(L.cstart[a])
            if q - (L.fullsec[c][a] + L.offA[c][kI]) >= L.cntI[c][kI]:
                continue                                                  # pad slot
            tot = 0.0
            for si, sg in enumerate(supers):
                if sg["c"] == c and kap < L.ke(a, sg["io"]):              # the super's rows reach k
                    tot += ypart[si][q]
            Jt[L.cbase[c] + q] = tot
    assert not np.isnan(Jt).any(), "the Jt reduction read a slot that no task wrote"
    D = np.zeros((N, N))                                                  # internal indices
    for x in range(N):
        xo = int(orig[x])
        for y in range(N):
            wy = int(L.chunk_of[y]); cy = int(L.clsI[y]); ly = int(loc[orig[y]])
            s = 0.0
            for gi in gfirst.get(x, []):                                  # x is the first index of the rows of these groups
                g = groups[gi]; c = g["c"]
                if L.task_exists(c, wy, xo):
                    s += DIc[gi, y]
                bcl = cy ^ c
                for w in range(L.wfirst[bcl], L.wfirst[bcl + 1]):
                    if L.kap0[c][w] <= ly < L.ke(cy, xo):
                        s += DIr[gi, L.rpoff[c][w] + ly]
            for z in range(N):                                            # rows (z, x): x is the second index
                zo = int(orig[z])
                if zo <= xo:
                    continue
                r = rowmap.get(L.key(z, x))
                if r is None:
                    continue
                c = int(L.clsI[z] ^ L.clsI[x])
                if L.task_exists(c, wy, zo):
                    s += DJc[r, y]
                bcl = cy ^ c
                for w in range(L.wfirst[bcl], L.wfirst[bcl + 1]):
                    if L.kap0[c][w] <= ly < L.ke(cy, zo):
                        s += DJr[r, L.rpoff[c][w] + ly]
            D[x, y] = s
    assert not np.isnan(D).any(), "a reduction read a partial that no task wrote"
    J = np.zeros((N, N)); K = np.zeros((N, N))
    for a_ in range(N):
        for b_ in range(N):
            sa, sb = int(sigma[a_]), int(sigma[b_])
            K[a_, b_] = D[sa, sb] + D[sb, sa]
            hi, lo = max(a_, b_), min(a_, b_)
            c = int(cls[hi] ^ cls[lo])
            q = int(L.cbase[c]) + L.pair_index(c, int(sigma[hi]), int(loc[lo]))
            s = Jt[q]
            r = rowmap.get(L.key(sa, sb))
            if r is not None:
                for w in range(L.NW):
                    if L.task_exists(c, w, hi):
                        s += Jd[r, w]
            J[a_, b_] = s
    assert not np.isnan(J).any()
    return J, K, dict(n_tasks=n_tasks, n_groups=len(groups), n_supers=len(supers))


def stored_elements(L: Layout):
    """padded doubles of the whole tensor (all rows i >= j)"""
    tot = 0
    for i in range(L.N):
        for j in range(i + 1):
            tot += L.row_len(i, j)
    return tot


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
    E = np.zeros((N, N, N, N))
    pidx = np.zeros((N, N), dtype=np.int64)
    pidx[ii, jj] = np.arange(npair); pidx[jj, ii] = np.arange(npair)
    E = A[pidx[:, :, None, None], pidx[None, None, :, :]]
    return E
