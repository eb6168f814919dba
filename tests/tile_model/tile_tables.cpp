// TEST INFRASTRUCTURE: exposes the host table builder of the "tiles" tensor layout (tuna_amd/csrc/tf_tiles_host.h) to the NumPy
// model tests/tile_model.py.  Built by tests/tile_model/build.sh with g++ (no HIP, no GPU); nothing in the product links it.
#include <cstring>
#include "../../tuna_amd/csrc/tf_tiles_host.h"

struct Handle {
    tft::ClassInfo C;
    tft::Tables T;
    tft::TaskList sub;          // a list of another strip height (ksub != 64), over the same regions
    std::vector<int> sigma;     // original -> internal
    std::string err;
};

extern "C" {

void *ttm_build(int N, const int *cls, int n_rows, const int *rows_ij /* original (i >= j) pairs */, int ksub, int part_steps)
{
    Handle *h = new Handle();
    tft::ClassInfo &C = h->C;
    C.N = N;
    for (int k = 0; k < N; ++k) ++C.csize[cls[k]];
    int order[4] = {0, 1, 2, 3};
    std::stable_sort(order, order + 4, [&](int x, int y) { return C.csize[x] > C.csize[y]; });
    for (int t = 0, s0 = 0; t < 4; ++t) { C.cstart[order[t]] = s0; s0 += C.csize[order[t]]; }
    h->sigma.assign(N, 0); C.clsI.assign(N, 0); C.origI.assign(N, 0); C.cntA.assign((size_t)4 * N, 0);
    std::vector<int> cnt((size_t)4 * N, 0);
    int seen[4] = {0, 0, 0, 0};
    for (int k = 0; k < N; ++k) {
        const int loc = seen[cls[k]]++;
        h->sigma[k] = C.cstart[cls[k]] + loc;
        C.origI[h->sigma[k]] = k; C.clsI[h->sigma[k]] = cls[k];
        for (int b = 0; b < 4; ++b) cnt[(size_t)b * N + k] = seen[b];
    }
    for (int a = 0; a < 4; ++a)
        for (int x = 0; x < N; ++x) C.cntA[(size_t)a * N + x] = cnt[(size_t)a * N + C.origI[x]];
    std::vector<std::pair<int, int>> rows;
    for (int r = 0; r < n_rows; ++r) rows.push_back({h->sigma[rows_ij[2 * r]], h->sigma[rows_ij[2 * r + 1]]});
    std::sort(rows.begin(), rows.end());
    h->err = tft::build(C, rows, part_steps, h->T);
    if (h->err.empty() && ksub != TT_KS) h->err = tft::build_list(C, h->T, rows, ksub, part_steps, h->sub);
    return h;
}
void ttm_free(void *p) { delete (Handle *)p; }
const char *ttm_error(void *p) { return ((Handle *)p)->err.c_str(); }
static tft::TaskList &list_of(Handle *h, int which) { return which ? h->sub : h->T.primary; }
// which: 0 primary list (ksub 64), 1 the sub list
long long ttm_count(void *p, int which, int what)
{
    Handle *h = (Handle *)p;
    tft::TaskList &L = list_of(h, which);
    switch (what) {
    case 0: return (long long)L.tasks.size();
    case 1: return h->T.n_elems;
    case 2: return h->T.edge_base;
    case 3: return L.dj_len;
    case 4: return L.jd_len;
    case 5: return L.jt_len;
    case 6: return L.n_di;
    case 7: return h->T.npair;
    case 8: return (long long)h->T.primary.tasks_by_region.size();
    case 9: return (long long)sizeof(TTask);
    case 10: return (long long)sizeof(TPairI);
    case 11: return (long long)sizeof(TRunI);
    case 12: return (long long)h->T.jlist.size();
    }
    return -1;
}
void ttm_copy(void *p, int which, int what, void *out)
{
    Handle *h = (Handle *)p;
    tft::TaskList &L = list_of(h, which);
    const int N = h->C.N;
    switch (what) {
    case 0: memcpy(out, L.tasks.data(), L.tasks.size() * sizeof(TTask)); break;
    case 1: memcpy(out, L.pairs.data(), L.pairs.size() * sizeof(TPairI)); break;
    case 2: memcpy(out, L.runs.data(), L.runs.size() * sizeof(TRunI)); break;
    case 3: memcpy(out, h->T.primary.tasks_by_region.data(), h->T.primary.tasks_by_region.size() * sizeof(TTask)); break;
    case 4: memcpy(out, h->sigma.data(), N * sizeof(int)); break;
    case 5: memcpy(out, h->C.origI.data(), N * sizeof(int)); break;
    case 6: memcpy(out, h->C.clsI.data(), N * sizeof(int)); break;
    case 7: memcpy(out, h->C.cstart, 4 * sizeof(int)); break;
    case 8: memcpy(out, h->T.pa, 10 * sizeof(int)); break;
    case 9: memcpy(out, h->T.pb, 10 * sizeof(int)); break;
    case 10: memcpy(out, L.itask_ptr.data(), (N + 1) * sizeof(int)); break;
    case 11: memcpy(out, L.itasks.data(), L.itasks.size() * sizeof(int)); break;
    case 12: memcpy(out, h->T.jlist_ptr.data(), (N + 1) * sizeof(int)); break;
    case 13: memcpy(out, h->T.jlist.data(), h->T.jlist.size() * sizeof(int)); break;
    case 14: memcpy(out, L.bucket, (TT_W + 1) * sizeof(int)); break;
    case 15: memcpy(out, h->C.cntA.data(), (size_t)4 * N * sizeof(int)); break;
    }
}
// the shape functions of tf_tiles.h, for the model
int ttm_row_len(int tri, int ks, int lb, int r, int nl) { return tt_row_len(tri != 0, ks, lb, r, nl); }
int ttm_row_off(int tri, int ks, int lb, int r, int nl) { return tt_row_off(tri != 0, ks, lb, r, nl); }
int ttm_elem_off(int tri, int ks, int lb, int r, int c, int nks, int nl) { return tt_elem_off(tri != 0, ks, lb, r, c, nks, nl); }
int ttm_dj_koff(int tri, int ks, int nl) { return tt_dj_koff(tri != 0, ks, nl); }
int ttm_dj_loff(int tri, int lb, int nk, int ksub) { return tt_dj_loff(tri != 0, lb, nk, ksub); }
int ttm_dj_first_sub(int tri, int lb, int ksub) { return tt_dj_first_sub(tri != 0, lb, ksub); }
int ttm_nlb(int tri, int ks, int nk, int nl) { return tt_nlb(tri != 0, ks, nk, nl); }
void ttm_chunks(int nlb, int *nch, int *w) { tt_chunks(nlb, nch, w); }
}

// address of the canonical element (internal indices) through the library's own address function (tf_tiles.h: tt_elem_addr)
extern "C" long long ttm_elem_addr(void *p, int iI, int jI, int kI, int lI)
{
    Handle *h = (Handle *)p;
    std::vector<int> tab(TVT_LEN, 0);
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) tab[TVT_PID + 4 * a + b] = std::max(0, h->T.pid[a][b]);
    for (int q = 0; q < h->T.npair; ++q) { tab[TVT_PA + q] = h->T.pa[q]; tab[TVT_PB + q] = h->T.pb[q]; tab[TVT_PMOFF + q] = h->T.pm_off[q]; tab[TVT_PMPITCH + q] = h->T.pm_pitch[q]; }
    for (int q = 0; q < 4; ++q) { tab[TVT_CSTART + q] = h->C.cstart[q]; tab[TVT_CSIZE + q] = h->C.csize[q]; }
    TView V{};
    V.regions = h->T.primary.tasks_by_region.data(); V.prim_pairs = h->T.primary.pairs.data(); V.prim_runs = h->T.primary.runs.data();
    V.edge_base = h->T.edge_base; V.N = h->C.N; V.pm_len = h->T.pm_len; V.tab = tab.data();
    return tt_elem_addr(V, h->C.clsI.data(), iI, jI, kI, lI);
}
