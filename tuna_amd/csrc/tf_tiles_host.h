// tf_tiles_host.h -- host tables of the "tiles" tensor layout (tf_tiles.h): regions of the stored tensor, task lists of the Fock
// kernel, shapes of the partial sums.  Pure C++ (no HIP): compiled into libtunafock (tf_device.hip) and into the CPU test library of
// tests/tile_model (the NumPy model of the Fock-build algebra runs on THESE tables).
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>
#include "tf_tiles.h"

namespace tft {

struct ClassInfo {               // parity classes of the output AOs (tf_device.hip: build_blocked_layout)
    int N = 0;
    int cstart[4] = {0, 0, 0, 0}, csize[4] = {0, 0, 0, 0};
    std::vector<int> clsI, origI;     // [N] by internal index: class, original index
    std::vector<int> cntA;            // [4][N]: class-a AOs with original index <= that of internal x
    int loc(int xI) const { return xI - cstart[clsI[xI]]; }
    int below(int a, int iI) const { return cntA[(size_t)a * N + iI] - (a == clsI[iI] ? 1 : 0); }   // class-a AOs strictly below i
};

struct TaskList {                // tasks of one strip height (ksub) and the shapes of the partial sums that go with it
    int ksub = TT_KS;
    std::vector<TTask> tasks;         // launch order: by waves (4, 3, 2, 1), heaviest first
    std::vector<TTask> tasks_by_region;   // primary list only: creation order = the REGIONS of the stored tensor (TPairI::first_task indexes it)
    int bucket[TT_W + 1] = {0, 0, 0, 0, 0};   // tasks with TT_W - b waves: [bucket[b], bucket[b + 1])
    std::vector<TPairI> pairs;        // [N][10]
    std::vector<TRunI> runs;          // [N][4]
    std::vector<int> itask_ptr, itasks;   // tasks by first index i (gather of the per-task outputs)
    long long dj_len = 0, jd_len = 0, jt_len = 0;
    int n_di = 0;                     // task waves
};

struct Tables {
    int N = 0;
    int pa[10], pb[10], npair = 0;    // class pairs (rows a, columns b); a == b: triangle
    int pid[4][4];                    // pair id of (a, b) in either order
    long long n_elems = 0;            // doubles of the stored tensor (interior regions, then the edge elements)
    long long edge_base = 0;
    long long max_slice = 0;
    int pm_off[10], pm_pitch[10], pm_len = 0;   // pair matrices: block of pair p = [csize[a]][pad2(csize[b])]
    TaskList primary;                 // ksub = 64: its tasks are also the REGIONS of the stored tensor (TPairI::first_task)
    std::vector<int> jlist_ptr, jlist;    // per second index j (internal): the first indices i != j of the owned rows (i, j), ascending
};

inline void class_pairs(const ClassInfo &C, Tables &T)
{
    T.npair = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) T.pid[a][b] = -1;
    for (int c = 0; c < 4; ++c)
        for (int a = 0; a < 4; ++a) {
            const int b = a ^ c;
            if (c != 0 && !(C.cstart[a] < C.cstart[b] || (C.cstart[a] == C.cstart[b] && a < b))) continue;   // rows: the class that comes first (the larger one)
            T.pa[T.npair] = a; T.pb[T.npair] = b;
            T.pid[a][b] = T.pid[b][a] = T.npair++;
        }
    T.pm_len = 0;
    for (int p = 0; p < 10; ++p) { T.pm_off[p] = 0; T.pm_pitch[p] = 0; }
    for (int p = 0; p < T.npair; ++p) {
        T.pm_off[p] = T.pm_len; T.pm_pitch[p] = tt_pad2(C.csize[T.pb[p]]);
        T.pm_len += C.csize[T.pa[p]] * T.pm_pitch[p];
    }
}

// Task list of strip height ksub over the regions of `T.primary` (ksub == 64: builds the regions themselves).
// rows: the owned rows as (internal i, internal j), sorted by (i, j).  part_steps: longest run of j steps of one task.
inline std::string build_list(const ClassInfo &C, Tables &T, const std::vector<std::pair<int, int>> &rows, int ksub, int part_steps, TaskList &L)
{
    const int N = C.N;
    const bool primary = (&L == &T.primary);
    L.ksub = ksub;
    L.tasks.clear();
    L.pairs.assign((size_t)N * 10, TPairI{});
    L.runs.assign((size_t)N * 4, TRunI{});
    for (auto &p : L.pairs) p.first_task = -1;
    // runs of owned j per (i, class of j)
    {
        size_t r = 0;
        while (r < rows.size()) {
            const int iI = rows[r].first, cj = C.clsI[rows[r].second];
            size_t e = r + 1;
            while (e < rows.size() && rows[e].first == iI && C.clsI[rows[e].second] == cj) {
                if (rows[e].second != rows[e - 1].second + 1) return "tiles layout: the owned rows of one first index and class are not a contiguous run";
                ++e;
            }
            TRunI &R = L.runs[(size_t)iI * 4 + cj];
            if (R.nj) return "tiles layout: several runs of one class under one first index";
            R.j0 = rows[r].second; R.nj = (int)(e - r);
            r = e;
        }
    }
    long long tensor_off = 0, jt_off = 0, dj_off = 0, jd_off = 0, edge_off = 0;
    int di_off = 0;
    struct Ord { int nw; long long work; int idx; };
    std::vector<Ord> ord;
    std::vector<TTask> tasks;
    for (int iI = 0; iI < N; ++iI) {
        const int ci = C.clsI[iI];
        // DJ vectors of the runs of this i: per row class c the pairs with a ^ b == c, in pair order: [K part | L part] each
        for (int cj = 0; cj < 4; ++cj) {
            TRunI &R = L.runs[(size_t)iI * 4 + cj];
            if (!R.nj) continue;
            const int c = ci ^ cj;
            int len = 0;
            for (int p = 0; p < T.npair; ++p) {
                const int a = T.pa[p], b = T.pb[p];
                if ((a ^ b) != c) continue;
                const bool tri = a == b;
                const int nk = C.below(a, iI), nl = tri ? nk : C.below(b, iI);
                TPairI &P = L.pairs[(size_t)iI * 10 + p];
                P.nk = nk; P.nl = nl; P.j0 = R.j0; P.nj = R.nj;
                if (nk <= 0 || nl <= 0) continue;
                P.dj_k = len; len += tt_dj_klen(tri, nk, nl);
                P.dj_l = len; len += tt_dj_llen(tri, nk, nl, ksub);
            }
            R.dj_len = len;
            R.dj_base = dj_off;
            dj_off += (long long)len * R.nj;
            const int l0 = R.j0 - C.cstart[cj];
            R.e_base = edge_off;                                       // (relative to Tables::edge_base)
            edge_off += (long long)(l0 + R.nj) * (l0 + R.nj + 1) / 2 - (long long)l0 * (l0 + 1) / 2;
        }
        for (int p = 0; p < T.npair; ++p) {
            const int a = T.pa[p], b = T.pb[p], c = a ^ b, cj = ci ^ c;
            const bool tri = a == b;
            const TRunI &R = L.runs[(size_t)iI * 4 + cj];
            TPairI &P = L.pairs[(size_t)iI * 10 + p];
            if (!R.nj || P.nk <= 0 || P.nl <= 0) continue;
            const int nk = P.nk, nl = P.nl;
            if (primary) {
                P.nparts = std::max(1, (R.nj + part_steps - 1) / part_steps);
                P.pj = (R.nj + P.nparts - 1) / P.nparts;
                P.nparts = (R.nj + P.pj - 1) / P.pj;
            } else {
                const TPairI &Q = T.primary.pairs[(size_t)iI * 10 + p];
                P.nparts = Q.nparts; P.pj = Q.pj;
            }
            P.jt_pitch = tt_pad2(nl);
            P.jt_base = jt_off;
            P.jt_part_stride = (long long)nk * P.jt_pitch;
            jt_off += P.jt_part_stride * P.nparts;
            P.first_task = primary ? (int)tasks.size() : T.primary.pairs[(size_t)iI * 10 + p].first_task;
            const int nstored = (nk + TT_KS - 1) / TT_KS;
            int per_part = 0;
            for (int part = 0; part < P.nparts; ++part) {
                const int s0 = part * P.pj, ns = std::min(P.pj, R.nj - s0);
                int tcount = 0;
                for (int ks = 0; ks < nstored; ++ks) {
                    const int nks_st = std::min(TT_KS, nk - TT_KS * ks);
                    int nch, w;
                    const int nlb = tt_nlb(tri, ks, nk, nl);
                    tt_chunks(nlb, &nch, &w);
                    for (int ch = 0; ch < nch; ++ch, ++tcount) {
                        const int lb0 = ch * w, nwst = std::min(w, nlb - lb0);
                        // the stored region of (part, strip, chunk)
                        long long base; int slice; int woff[TT_W] = {0, 0, 0, 0};
                        if (primary) {
                            int off = 0;
                            for (int u = 0; u < nwst; ++u) { woff[u] = off; off += tt_piece_len(tri, ks, lb0 + u, nks_st, nl); }
                            slice = off;
                            base = tensor_off;
                            tensor_off += (long long)slice * ns;
                            T.max_slice = std::max<long long>(T.max_slice, slice);
                        } else {
                            const TTask &Rg = T.primary.tasks_by_region[(size_t)P.first_task + (size_t)part * T.primary.pairs[(size_t)iI * 10 + p].tasks_per_part + tcount];
                            base = Rg.base; slice = Rg.slice;
                            for (int u = 0; u < TT_W; ++u) woff[u] = Rg.woff[u];
                        }
                        // the tasks of the region: its sub-strips of ksub rows
                        for (int r0 = 0; r0 < nks_st; r0 += ksub) {
                            const int k0 = TT_KS * ks + r0, nks = std::min(ksub, nks_st - r0);
                            int nw = nwst;
                            if (tri) {                                  // blocks beyond the sub-strip's last row hold nothing for it
                                const int reach = (k0 + nks - 1) / TT_LB + 1;
                                nw = std::min(nwst, reach - lb0);
                                if (nw <= 0) continue;
                            }
                            TTask t{};
                            t.base = base; t.slice = slice;
                            for (int u = 0; u < TT_W; ++u) t.woff[u] = woff[u];
                            t.i = iI; t.j0 = R.j0 + s0; t.nj = ns; t.a = a; t.b = b; t.k0 = k0; t.nks = nks; t.roff0 = r0;
                            t.lb0 = lb0; t.nw = nw; t.nk = nk; t.nl = nl; t.pid = p;
                            t.kbase = C.cstart[a]; t.lbase = C.cstart[b]; t.ncol = C.csize[b]; t.pm_off = T.pm_off[p]; t.pm_pitch = T.pm_pitch[p];
                            t.jt_base = P.jt_base + (long long)part * P.jt_part_stride; t.jt_pitch = P.jt_pitch;
                            t.dj_len = R.dj_len; t.dj_base = R.dj_base + (long long)s0 * R.dj_len;
                            t.dj_koff = P.dj_k + tt_dj_koff(tri, ks, nl) + ch * nks_st + r0;
                            const int sub = k0 / ksub;
                            for (int u = 0; u < nw; ++u) {
                                const int lb = lb0 + u;
                                t.dj_loff[u] = P.dj_l + tt_dj_loff(tri, lb, nk, ksub) + (sub - tt_dj_first_sub(tri, lb, ksub)) * TT_LB;
                            }
                            t.di_base = di_off; di_off += nw;
                            t.jd_base = (int)jd_off; jd_off += (long long)nw * ns;
                            t.self_last = (c == 0 && t.j0 + t.nj - 1 == iI) ? 1 : 0;
                            ord.push_back(Ord{nw, (long long)ns * slice * nks / std::max(1, nks_st), (int)tasks.size()});
                            tasks.push_back(t);
                        }
                    }
                }
                per_part = tcount;
            }
            P.tasks_per_part = per_part;
        }
    }
    if (jd_off > 0x7fffffffLL) return "tiles layout: too many per-step partial sums for 32-bit offsets";
    if (primary) { T.edge_base = tensor_off; T.n_elems = tensor_off + edge_off; T.primary.tasks_by_region = tasks; }
    L.dj_len = dj_off; L.jd_len = jd_off; L.jt_len = jt_off; L.n_di = di_off;
    // launch order
    std::stable_sort(ord.begin(), ord.end(), [](const Ord &x, const Ord &y) { return x.nw != y.nw ? x.nw > y.nw : x.work > y.work; });
    L.tasks.resize(tasks.size());
    std::vector<int> newpos(tasks.size());
    for (int b = 0; b <= TT_W; ++b) L.bucket[b] = (int)tasks.size();
    for (size_t k = 0; k < ord.size(); ++k) {
        L.tasks[k] = tasks[(size_t)ord[k].idx];
        newpos[(size_t)ord[k].idx] = (int)k;
        const int b = TT_W - ord[k].nw;
        if ((int)k < L.bucket[b]) L.bucket[b] = (int)k;
    }
    for (int b = TT_W - 1; b >= 0; --b) L.bucket[b] = std::min(L.bucket[b], L.bucket[b + 1]);
    L.bucket[0] = 0;
    // tasks by first index (creation order inside an i: a fixed summation order for the gather)
    L.itask_ptr.assign((size_t)N + 1, 0);
    for (const TTask &t : tasks) ++L.itask_ptr[(size_t)t.i + 1];
    for (int x = 0; x < N; ++x) L.itask_ptr[(size_t)x + 1] += L.itask_ptr[(size_t)x];
    L.itasks.assign(tasks.size(), 0);
    {
        std::vector<int> fill(L.itask_ptr.begin(), L.itask_ptr.end() - 1);
        for (size_t k = 0; k < tasks.size(); ++k) L.itasks[(size_t)fill[(size_t)tasks[k].i]++] = newpos[k];
    }
    return "";
}

inline std::string build(const ClassInfo &C, const std::vector<std::pair<int, int>> &rows, int part_steps, Tables &T)
{
    T = Tables();
    T.N = C.N;
    class_pairs(C, T);
    std::string e = build_list(C, T, rows, TT_KS, part_steps, T.primary);
    if (!e.empty()) return e;
    T.jlist_ptr.assign((size_t)C.N + 1, 0);
    for (const auto &r : rows) if (r.first != r.second) ++T.jlist_ptr[(size_t)r.second + 1];
    for (int x = 0; x < C.N; ++x) T.jlist_ptr[(size_t)x + 1] += T.jlist_ptr[(size_t)x];
    T.jlist.assign((size_t)std::max(1, T.jlist_ptr[(size_t)C.N]), 0);
    std::vector<int> fill(T.jlist_ptr.begin(), T.jlist_ptr.end() - 1);
    for (const auto &r : rows) if (r.first != r.second) T.jlist[(size_t)fill[(size_t)r.second]++] = r.first;
    return "";
}

}  // namespace tft
