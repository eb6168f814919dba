// tf_tiles.h -- the "tiles" tensor layout: shapes shared by the host table builder (tf_tiles_host.h), the writer
// (xform_bra_store_tiles), the Fock-build kernels (tf_jktile.hip.h) and the CPU model of tests/tile_model.py.
// Reference: the tensor of calculate_electron_repulsion_integrals (pyx:1267-1355) as consumed by calculate_coulomb_matrix
// (scf:55-72) and calculate_exchange_matrix (scf:27-44).
//
// What is stored: every symmetry-unique, parity-allowed (ij|kl), once -- canonical form i >= j, k >= l, (kl) <= (ij) in ORIGINAL AO
// order (what the generation produces), class(i) ^ class(j) == class(k) ^ class(l) (x/y reflection parity, pyx:1324-1327).
//   * INTERIOR (k < i): for a first index i and an unordered pair of parity classes {a, b} the elements with k of class a, l of
//     class b, both below i, form a RECTANGLE [k][l] (a != b: every unordered pair {k, l} exactly once) or a TRIANGLE l <= k
//     (a == b) whose shape does not depend on j.  They are stored j-innermost: the block is cut into strips of TT_KS rows k and
//     blocks of TT_LB columns l; a TASK = (i, class pair, strip, chunk of <= TT_W column blocks, run of j) owns one contiguous region
//     [j][column block][row k][l], so that one step of the Fock kernel -- one j -- reads one contiguous slice, each wave its piece:
//     rows of <= 16 values (a full row: one 128-byte line), padded to an even count; triangles store l <= k only.
//   * EDGE (k == i, l <= j): E[i][j][l], a triangle per class of j; 0.3 % of the tensor, handled by jk_edge_kernel.
// Internal AO index sigma = cstart[class] + loc (classes sorted by size, original order inside a class), as in tf_layout.hip.h.
#pragma once

#if defined(__HIPCC__)
#define TT_HD __host__ __device__ __forceinline__
#else
#define TT_HD inline
#endif

#define TT_KS 64            // rows of a strip in memory (4 MFMA row blocks of 16)
#define TT_LB 16            // columns of a block (one wave: 4 K-steps of v_mfma_f64_16x16x4)
#define TT_W 4              // column blocks (waves) of a workgroup at most
#define TT_KB 4             // steps between two merges of the row sums in LDS

TT_HD int tt_min(int a, int b) { return a < b ? a : b; }
TT_HD int tt_max(int a, int b) { return a > b ? a : b; }
TT_HD int tt_pad2(int n) { return (n + 1) & ~1; }

// column blocks that reach strip ks (rows 64 ks .. of nk): all of them (rectangle) or those up to the strip's last row (triangle)
TT_HD int tt_nlb(bool tri, int ks, int nk, int nl)
{
    if (!tri) return (nl + TT_LB - 1) / TT_LB;
    const int kmax = tt_min(TT_KS * ks + TT_KS - 1, nk - 1);
    return kmax / TT_LB + 1;
}
// the blocks of a strip are dealt to nch workgroups of w waves (the last one may have fewer)
TT_HD void tt_chunks(int nlb, int *nch, int *w)
{
    *nch = (nlb + TT_W - 1) / TT_W;
    *w = *nch ? (nlb + *nch - 1) / *nch : 0;
}
// triangle: padded length of the row whose diagonal distance from the block's first column is d (l <= k: d + 1 values, at most 16)
TT_HD int tt_tri_rowlen(int d) { return d < 0 ? 0 : tt_min(2 * (d >> 1) + 2, TT_LB); }
// sum of tt_tri_rowlen(d') over 0 <= d' < d
TT_HD int tt_triF(int d)
{
    if (d <= 0) return 0;
    if (d >= TT_LB) return 144 + TT_LB * (d - TT_LB);
    const int q = d >> 1;
    return 2 * q * (q + 1) + ((d & 1) ? 2 * q + 2 : 0);
}
// a wave's piece: strip ks (its rows r = 0 .. nks - 1 are k = 64 ks + r), column block lb.  nl: columns of the block's class below i.
TT_HD int tt_row_len(bool tri, int ks, int lb, int r, int nl)          // padded values of row r
{
    if (!tri) return tt_pad2(tt_min(TT_LB, nl - TT_LB * lb));
    return tt_tri_rowlen(TT_KS * ks + r - TT_LB * lb);
}
TT_HD int tt_row_off(bool tri, int ks, int lb, int r, int nl)          // start of row r inside the piece
{
    if (!tri) return r * tt_pad2(tt_min(TT_LB, nl - TT_LB * lb));
    const int d0 = TT_KS * ks - TT_LB * lb;
    return tt_triF(d0 + r) - tt_triF(d0);
}
// Inside a block of 16 rows the values are NOT row-major: the Fock kernel loads a piece straight into the A-operand layout of
// v_mfma_f64_16x16x4 (lane = 16 kk + m: row m, columns 4 kk + 2 h .. + 1 with load h), and a quarter wave -- 16 lanes, the unit the
// memory pipeline coalesces -- would touch 16 different lines of a row-major tile.  The block is therefore stored chunk by chunk: for
// h = 0, 1 and kk = 0 .. 3 the column pair c0 = 4 kk + 2 h of every row that has it (a triangle's first rows are shorter), row after
// row: what a quarter wave loads is one contiguous run (256 bytes of a full block, whose eight chunks are 2 KB in load order).  Same
// doubles as row-major.  r: row of the stored strip; c: column of the block (0 .. 15).
TT_HD int tt_elem_off(bool tri, int ks, int lb, int r, int c, int nks, int nl)
{
    const int r0 = r & ~15, m = r - r0, nr = tt_min(16, nks - r0);
    const int base = tt_row_off(tri, ks, lb, r0, nl);
    const int d0 = TT_KS * ks + r0 - TT_LB * lb;                       // triangle: diagonal distance of the block's first row
    const int rl = tri ? 0 : tt_pad2(tt_min(TT_LB, nl - TT_LB * lb)); // rectangle: padded row length
    const int c0 = c & ~1;
    int off = base;
    for (int h = 0; h < 2; ++h)
        for (int kq = 0; kq < 4; ++kq) {
            const int cc = 4 * kq + 2 * h;
            const int m0 = tri ? tt_max(0, cc - d0) : (cc < rl ? 0 : 16);   // first row of the block that has the pair cc
            if (cc == c0) return off + 2 * (m - m0) + (c - c0);
            off += 2 * tt_max(0, nr - m0);
        }
    return off;
}
TT_HD int tt_piece_len(bool tri, int ks, int lb, int nks, int nl)      // doubles of the piece, a multiple of 16 (128 bytes)
{
    return (tt_row_off(tri, ks, lb, nks, nl) + 15) & ~15;
}

// ---- per-row partial vectors of the exchange terms D[j][.] ("DJ parts") ------------------------------------------------------
// A stored row (i, j) receives D[j][k] += sum_l m P[i][l] from every workgroup (strip, chunk) and D[j][l] += sum_k m P[i][k] from
// every wave (strip, block).  Its vector holds, per class pair of its row class: the K part [stored strip][chunk][row of the strip]
// and the L part [block][sub-strip that reaches it][16].  `ksub` = rows of a TASK's strip: 64 for one density per pass; 32 / 16 when
// the register budget of several densities cuts the strips (the tasks then cover a part of the stored strip: disjoint rows of the
// same K slot, but one L slot each).
TT_HD int tt_dj_koff(bool tri, int ks, int nl)                          // start of stored strip ks in the K part (the strips before it are full)
{
    if (tri) return TT_KS * (ks * (ks + 1) / 2);                       // strip t < ks: 4 t + 4 blocks = t + 1 chunks
    int nch, w;
    tt_chunks((nl + TT_LB - 1) / TT_LB, &nch, &w);
    return nch * TT_KS * ks;
}
TT_HD int tt_dj_klen(bool tri, int nk, int nl)
{
    if (nk <= 0) return 0;
    const int ksl = (nk - 1) / TT_KS;
    int nch, w;
    tt_chunks(tt_nlb(tri, ksl, nk, nl), &nch, &w);
    return tt_dj_koff(tri, ksl, nl) + nch * (nk - TT_KS * ksl);
}
TT_HD int tt_dj_first_sub(bool tri, int lb, int ksub) { return tri ? (TT_LB * lb) / ksub : 0; }   // first sub-strip that reaches block lb
TT_HD int tt_dj_loff(bool tri, int lb, int nk, int ksub)               // start of block lb in the L part
{
    const int ns = (nk + ksub - 1) / ksub;
    if (!tri) return lb * ns * TT_LB;
    const int g = ksub / TT_LB, q = lb / g, rem = lb - q * g;          // sum over t < lb of floor(t / g)
    return TT_LB * (lb * ns - (g * (q * (q - 1) / 2) + rem * q));
}
TT_HD int tt_dj_llen(bool tri, int nk, int nl, int ksub)
{
    const int nlb = tri ? (nk + TT_LB - 1) / TT_LB : (nl + TT_LB - 1) / TT_LB;
    return tt_dj_loff(tri, nlb, nk, ksub);
}

// ---- tables ---------------------------------------------------------------------------------------------------------------------
struct TTask {                   // one workgroup of jk_tile_kernel
    long long base;              // first double of the task's region in the tensor (slice of step s: base + s * slice)
    long long jt_base;           // Jt partial block of (i, class pair, part): [k][pitch]
    long long dj_base;           // DJ vector of the row (i, j0) (step s: + s * dj_len)
    int i;                       // internal first index
    int j0, nj;                  // internal j of step 0; steps (consecutive internal j of one class)
    int a, b;                    // classes of k (rows) and l (columns); a == b: triangle
    int k0, nks;                 // loc of the task's first row, its rows (<= ksub)
    int roff0;                   // rows of the STORED strip in front of the task's first row (k0 - 64 (k0 / 64))
    int lb0, nw;                 // first column block, blocks (= waves)
    int nk, nl;                  // rows / columns of the pair's block below i (triangle: nl == nk)
    int slice;                   // doubles of one j slice of the stored (strip, chunk)
    int woff[TT_W];              // start of wave w's piece inside a slice
    int jt_pitch;
    int dj_len, dj_koff;         // DJ vector length; slot of this (sub-strip, chunk) in it
    int dj_loff[TT_W];           // slot of (block lb0 + w, this sub-strip)
    int di_base, jd_base;        // per-task outputs: DIk[(di_base + w) 64 ..], DIl[(di_base + w) 16 ..]; Jd[jd_base + w nj + s]
    int self_last;               // 1: the last step is j == i (its D[j][.] terms are dropped)
    int pid;                     // class pair id
    int kbase, lbase;            // internal index of the first AO of class a / class b
    int ncol;                    // AOs of class b (columns beyond read as zero)
    int pm_off, pm_pitch;        // the pair's block of the pair matrices (packed density, Jt totals): [k loc][pitch]
    int pad_;
};

struct TPairI {                  // (i, class pair): where the stored elements and the partial sums live
    int first_task;              // tasks of (i, pair): [part][strip][chunk] from here (-1: no owned row)
    int nparts, tasks_per_part;
    int j0, nj, pj;              // the owned run of j (internal) and the steps of a part
    int nk, nl;
    long long jt_base;           // Jt partial block of part 0 (part p: + p * jt_part_stride)
    long long jt_part_stride;
    int jt_pitch;
    int dj_k, dj_l;              // K part / L part of this pair inside a DJ vector
};

struct TRunI {                   // (i, class of j): the owned rows (i, j0 .. j0 + nj - 1)
    int j0, nj;                  // internal j (nj == 0: none)
    int dj_len, pad;
    long long dj_base;           // DJ vectors of the run
    long long e_base;            // edge elements E[i][j][l], l <= j: e_base + T(loc j) - T(loc j0) + loc l, T(n) = n (n + 1) / 2
};

// view of the tables of tf_tiles_host.h (device pointers in the kernels; host pointers in the CPU test library)
struct TView {
    const TTask *regions;         // tasks_by_region of the primary list
    const TPairI *prim_pairs;     // [N][10] of the primary list (regions of a pair)
    const TRunI *prim_runs;       // [N][4]
    long long edge_base;
    int N, pm_len;
    const int *tab;               // the small tables below (device memory: kernels index them with run-time class numbers, and a
                                  // dynamically indexed kernel argument would be copied to scratch memory)
};
// tab: pair id of two classes [4][4]; row / column class of a pair; internal range of a class; the pairs' blocks of the pair matrices
// (packed densities, Jt totals): block of pair p = [k loc][pitch]
enum { TVT_PID = 0, TVT_PA = 16, TVT_PB = 26, TVT_CSTART = 36, TVT_CSIZE = 40, TVT_PMOFF = 44, TVT_PMPITCH = 54, TVT_LEN = 64 };

// Address of the canonical element (ij|kl) -- internal indices; i >= j, k >= l, (kl) <= (ij) in ORIGINAL order -- or -1 (row not owned).
TT_HD long long tt_elem_addr(const TView &V, const int *clsI, int iI, int jI, int kI, int lI)
{
    const int cj = clsI[jI], ck = clsI[kI], cl = clsI[lI];
    const TRunI R = V.prim_runs[(size_t)iI * 4 + cj];
    if (jI < R.j0 || jI >= R.j0 + R.nj) return -1;
    if (kI == iI) {                                                       // edge: l <= j, of j's class
        const long long lj = jI - V.tab[TVT_CSTART + cj], l0 = R.j0 - V.tab[TVT_CSTART + cj];
        return V.edge_base + R.e_base + lj * (lj + 1) / 2 - l0 * (l0 + 1) / 2 + (lI - V.tab[TVT_CSTART + cl]);
    }
    const int p = V.tab[TVT_PID + ck * 4 + cl], a = V.tab[TVT_PA + p], b = V.tab[TVT_PB + p];
    const bool tri = a == b;
    int kr = kI - V.tab[TVT_CSTART + ck], lc = lI - V.tab[TVT_CSTART + cl];
    if (!tri && ck != a) { const int t = kr; kr = lc; lc = t; }
    const TPairI P = V.prim_pairs[(size_t)iI * 10 + p];
    if (P.first_task < 0) return -1;
    const int sj = jI - R.j0, part = sj / P.pj, s = sj - part * P.pj;
    const int ks = kr / TT_KS, lb = lc / TT_LB;
    int nch, w;
    tt_chunks(tt_nlb(tri, ks, P.nk, P.nl), &nch, &w);
    const int ch = lb / w, wv = lb - ch * w;
    const int before = tri ? ks * (ks + 1) / 2 : ks * nch;                 // chunks of the (full) strips in front
    const TTask *Rg = V.regions + (size_t)P.first_task + (size_t)part * P.tasks_per_part + before + ch;
    return Rg->base + (long long)s * Rg->slice + Rg->woff[wv] + tt_elem_off(tri, ks, lb, kr - TT_KS * ks, lc - TT_LB * lb, tt_min(TT_KS, P.nk - TT_KS * ks), P.nl);
}
