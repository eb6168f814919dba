// tf_jkpacked.hip.h -- Fock build from the 8-fold symmetry-unique tensor ("packed" layout).
// Reference: calculate_coulomb_matrix tuna_scf.py:55-72 ("ijkl,kl->ij"), calculate_exchange_matrix tuna_scf.py:27-44
// ("ilkj,kl->ij"); the reference keeps all 8 images of every (ij|kl) (pyx:1335-1342), here each unique value is stored once.
//
// Layout: pair index p = i(i+1)/2 + j (i >= j).  Row p of the tensor holds (ij|kl) for every pair q = k(k+1)/2 + l <= p,
// contiguously, i.e. the tensor is the packed lower triangle of the symmetric npair x npair matrix; a rank stores the rows it owns
// in ascending p with a row-offset table.  8 N^4 / 8 bytes instead of the reference's 8 N^4.
//
// One pass over row (i,j) has to feed six outputs per element m = (ij|kl):
//     Jd[ij] += m Pp[kl]                      (Pp[kl] = P[k][l] + P[l][k], or P[k][k])
//     Jt[kl] += m Pp[ij]                      (the transposed image (kl|ij); not for kl == ij)
//     D[i][k] += m P[j][l]   D[i][l] += m P[j][k] (k != l)     D[j][k] += m P[i][l] (i != j)    D[j][l] += m P[i][k] (i != j, k != l)
// (the four D terms at half weight when kl == ij), and K = D + D^T covers the transposed images when P is symmetric, as every SCF
// density is.  A general P takes two passes: K = D(P^T) + D(P)^T (the einsum of scf:42 exactly).
//
// Kernel shape: a workgroup owns up to JBB rows (i; j0..j0+nr-1) sharing i; a wave owns two 64-column chunks of every triangle
// row k (chunk c and its mirror NW-1-c, so all waves do the same work) and walks k up to i.  Lane-local accumulators: the
// "column" sums (outputs indexed by l); the "row" sums (outputs indexed by k) are reduced across the wave with a transposing
// butterfly on permlane swaps / DPP (no LDS) and written per wave.  Jt partials are written once per workgroup and column
// (1/JBB of the tensor's bytes) and summed by jt_reduce_kernel.  No atomics anywhere: results are bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>

#define TF_JKP_JBB 8
#define TF_JKP_SEG 64            // segments of the group list in the Jt reduction

struct JKGroup {
    int i, j0, nr, r0;           // rows r0..r0+nr-1 (local numbering) = pairs (i, j0..j0+nr-1)
    long long yoff;              // first element of this group's Jt partial (length = pair(i, j0+nr-1) + 1)
    long long ylen;
};

__device__ __forceinline__ double ld_stream(const double *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(double *p, double v) { __builtin_nontemporal_store(v, p); }

// ---- wave-level sums without LDS traffic -----------------------------------------------------------------------------
// gfx950's v_permlane32_swap / v_permlane16_swap exchange half-waves / odd-even 16-lane rows between two registers: one
// "transposing" butterfly step on a pair of values (a, b) costs two swaps and one add and leaves the pair-sums of a in one half and
// those of b in the other.  Distances < 16 use DPP row operations.
typedef unsigned tf_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double pair_step32(double a, double b)      // lanes 0-31: a[l] + a[l+32];  lanes 32-63: b[l-32] + b[l]
{
    const tf_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const tf_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}

__device__ __forceinline__ double pair_step16(double a, double b)      // even 16-lane rows: sums of a;  odd rows: sums of b
{
    const tf_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const tf_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}

template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_merge(double old, double src)    // lanes of the banks in BANK: src permuted by CTRL; others: old
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK, false);
    return __hiloint2double(hi, lo);
}

#define TF_DPP_ROR8 0x128
#define TF_DPP_HALF_MIRROR 0x141
#define TF_DPP_QUAD_IDENT 0xE4
#define TF_DPP_QUAD_XOR1 0xB1      // quad_perm [1,0,3,2]
#define TF_DPP_QUAD_XOR2 0x4E      // quad_perm [2,3,0,1]

__device__ __forceinline__ double pair_step8(double a, double b)       // lanes 0-7 of every row: sums of a;  lanes 8-15: sums of b
{
    double recv = dpp_merge<TF_DPP_ROR8, 0x3>(a, a);                    // banks 0,1 <- a[lane + 8]
    recv = dpp_merge<TF_DPP_ROR8, 0xC>(recv, b);                        // banks 2,3 <- b[lane - 8]
    const double keep = dpp_merge<TF_DPP_QUAD_IDENT, 0xC>(a, b);        // banks 2,3 keep b
    return keep + recv;
}

__device__ __forceinline__ double sum8(double t)                        // all 8 lanes of an aligned group get the group's sum
{
    t += dpp_merge<TF_DPP_QUAD_XOR1, 0xF>(t, t);
    t += dpp_merge<TF_DPP_QUAD_XOR2, 0xF>(t, t);
    t += dpp_merge<TF_DPP_HALF_MIRROR, 0xF>(t, t);
    return t;
}

// eight per-lane values -> lane L holds the wave total of value L >> 3
__device__ __forceinline__ double wave_sum8(const double (&v)[8])
{
    const double w0 = pair_step32(v[0], v[4]), w1 = pair_step32(v[1], v[5]), w2 = pair_step32(v[2], v[6]), w3 = pair_step32(v[3], v[7]);
    const double u0 = pair_step16(w0, w2), u1 = pair_step16(w1, w3);
    return sum8(pair_step8(u0, u1));
}

__device__ __forceinline__ double wave_sum1(double t)                   // every lane gets the wave total
{
    t = pair_step32(t, t);
    t = pair_step16(t, t);
    t += dpp_merge<TF_DPP_ROR8, 0xF>(t, t);
    return sum8(t);
}

// X = P or P^T (dense [N][N], what the exchange terms contract with); Pp[k(k+1)/2+l] = P[k][l] + P[l][k] (k != l), P[k][k]
__global__ void pack_density_kernel(const double *__restrict__ P, int N, int transpose, double *__restrict__ X, double *__restrict__ Pp)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int k = e / N, l = e - k * N;
    const double a = P[e], b = P[(size_t)l * N + k];
    X[e] = transpose ? b : a;
    if (l <= k) Pp[(size_t)k * (k + 1) / 2 + l] = (k == l) ? a : a + b;
}

// Per-wave state of one 64-column chunk: lane-local column sums and the P values of the lane's column.
struct JKChunk {
    int l;                               // column of this lane
    double pil, pjl[TF_JKP_JBB];         // P[i][l], P[j_r][l]
    double colI, colJ[TF_JKP_JBB];       // D[i][l], D[j_r][l] accumulators
};

enum { JKP_OFF = 0, JKP_DIAG = 1, JKP_FULL = 2 };

// One triangle row k < i of one chunk.  MODE FULL: every lane has l < k (no masks); DIAG: the 64x64 tile on the diagonal.
template <int MODE>
__device__ __forceinline__ void jkp_chunk_row(JKChunk &C, const double *const (&Tr)[TF_JKP_JBB], const double *__restrict__ Pp,
                                              double *__restrict__ yrow, long long bk, int k, const double (&pjk)[TF_JKP_JBB], double pik,
                                              const double (&ppij)[TF_JKP_JBB], double (&jd)[TF_JKP_JBB], double (&rJ)[TF_JKP_JBB], double &rI)
{
    if (MODE == JKP_OFF) return;
    const bool v = (MODE == JKP_FULL) || C.l <= k;
    const double offd = (MODE == JKP_FULL || C.l < k) ? 1.0 : 0.0;
    double m[TF_JKP_JBB];
#pragma unroll
    for (int r = 0; r < TF_JKP_JBB; ++r) m[r] = v ? ld_stream(Tr[r] + bk + C.l) : 0.0;
    const double pp = v ? Pp[bk + C.l] : 0.0;
    double jt = 0.0;
#pragma unroll
    for (int r = 0; r < TF_JKP_JBB; ++r) {
        const double mr = m[r];
        jd[r] += mr * pp;
        jt += mr * ppij[r];
        rI += mr * C.pjl[r];
        rJ[r] += mr * C.pil;
        const double mc = (MODE == JKP_FULL) ? mr : mr * offd;
        C.colI += mc * pjk[r];
        C.colJ[r] += mc * pik;
    }
    if (v) st_stream(yrow + C.l, jt);
}

// The last triangle row k == i: row r ends at l == j_r, where the element (ij|ij) counts half in K and not at all in Jt.
__device__ __forceinline__ void jkp_chunk_last(JKChunk &C, const JKGroup &g, const double *const (&Tr)[TF_JKP_JBB],
                                               const double *__restrict__ Pp, double *__restrict__ yrow, long long bk,
                                               const double (&pjk)[TF_JKP_JBB], double pik, const double (&ppij)[TF_JKP_JBB],
                                               double (&jd)[TF_JKP_JBB], double (&rJ)[TF_JKP_JBB], double &rI)
{
    const int i = g.i, jlast = g.j0 + g.nr - 1;
    const bool v = C.l <= jlast;
    const double pp = v ? Pp[bk + C.l] : 0.0;
    double jt = 0.0;
#pragma unroll
    for (int r = 0; r < TF_JKP_JBB; ++r) {
        const int jr = g.j0 + r;
        const bool vr = r < g.nr && C.l <= jr;
        const double mr = vr ? ld_stream(Tr[r] + bk + C.l) : 0.0;
        const bool diag = (C.l == jr);
        jd[r] += mr * pp;
        jt += diag ? 0.0 : mr * ppij[r];
        const double mk = diag ? 0.5 * mr : mr;
        rI += mk * C.pjl[r];
        rJ[r] += mk * C.pil;
        const double mc = (C.l < i) ? mk : 0.0;
        C.colI += mc * pjk[r];
        C.colJ[r] += mc * pik;
    }
    if (v) st_stream(yrow + C.l, jt);
}

// Workgroup = W = ceil(NW / 2) waves (NW = ceil(N / 64) column chunks); wave w owns chunks cA = w and cB = NW - 1 - w, which
// balances the triangle (chunk c is only populated for k >= 64 c).  After the P rows are staged in LDS the waves never meet again.
// Outputs: Jd [n_rows][W] per-wave partials; ypart: Jt partials; DIc [G][N], DJc [n_rows][N]: column parts (l-indexed);
// DIr [G][W][N], DJr [n_rows][W][N]: row parts per wave (k-indexed, written for 64 w <= k <= i).
__global__ __launch_bounds__(512) void jk_packed_kernel(const double *__restrict__ T, const long long *__restrict__ rowoff,
                                                        const JKGroup *__restrict__ groups, int N, const double *__restrict__ P,
                                                        const double *__restrict__ Pp, double *__restrict__ Jd,
                                                        double *__restrict__ ypart, double *__restrict__ DIc, double *__restrict__ DIr,
                                                        double *__restrict__ DJc, double *__restrict__ DJr)
{
    constexpr int JBB = TF_JKP_JBB;
    extern __shared__ double smem[];
    const JKGroup g = groups[blockIdx.x];
    const int W = blockDim.x >> 6, NW = (N + 63) >> 6;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // w in an SGPR: uniform loop control
    const int i = g.i;
    double *sPi = smem;                                  // P[i][.]
    double *sPj = smem + N;                              // [JBB][N]  P[j_r][.]
    for (int x = threadIdx.x; x < N; x += blockDim.x) {
        sPi[x] = P[(size_t)i * N + x];
#pragma unroll
        for (int r = 0; r < JBB; ++r) sPj[r * N + x] = (r < g.nr) ? P[(size_t)(g.j0 + r) * N + x] : 0.0;
    }
    __syncthreads();

    const int cA = w, cB = NW - 1 - w;
    const bool haveB = cB > cA;
    JKChunk A, B;
    A.l = cA * 64 + lane; B.l = cB * 64 + lane;
    {
        const bool inA = A.l < N, inB = haveB && B.l < N;
        A.pil = inA ? sPi[A.l] : 0.0; B.pil = inB ? sPi[B.l] : 0.0;
        A.colI = 0.0; B.colI = 0.0;
#pragma unroll
        for (int r = 0; r < JBB; ++r) {
            A.pjl[r] = inA ? sPj[r * N + A.l] : 0.0; B.pjl[r] = inB ? sPj[r * N + B.l] : 0.0;
            A.colJ[r] = 0.0; B.colJ[r] = 0.0;
        }
    }
    double ppij[JBB], jd[JBB];
    const double *Tr[JBB];
#pragma unroll
    for (int r = 0; r < JBB; ++r) {
        const bool have = r < g.nr;                      // missing rows alias row 0 with zero weights everywhere
        ppij[r] = have ? Pp[(size_t)i * (i + 1) / 2 + g.j0 + r] : 0.0;
        Tr[r] = T + rowoff[g.r0 + (have ? r : 0)];
        jd[r] = 0.0;
    }
    double *yg = ypart + g.yoff;
    double *DIr_w = DIr + ((size_t)blockIdx.x * W + w) * N;
    const size_t rowW = (size_t)W * N;
    double *DJr_w = DJr + ((size_t)g.r0 * W + w) * N;     // + r * rowW

    // one k-row of the wave: both chunks, then the row sums
#define JKP_ROW(MA, MB)                                                                                                        \
    {                                                                                                                          \
        const long long bk = (long long)k * (k + 1) / 2;                                                                       \
        double pjk[JBB], rJ[JBB], rI = 0.0;                                                                                    \
        _Pragma("unroll") for (int r = 0; r < JBB; ++r) { pjk[r] = sPj[r * N + k]; rJ[r] = 0.0; }                            \
        const double pik = sPi[k];                                                                                             \
        jkp_chunk_row<MA>(A, Tr, Pp, yg + bk, bk, k, pjk, pik, ppij, jd, rJ, rI);                                              \
        jkp_chunk_row<MB>(B, Tr, Pp, yg + bk, bk, k, pjk, pik, ppij, jd, rJ, rI);                                              \
        const double tJ = wave_sum8(rJ);                                                                                       \
        const double tI = wave_sum1(rI);                                                                                       \
        if ((lane & 7) == 0 && (lane >> 3) < g.nr) DJr_w[(size_t)(lane >> 3) * rowW + k] = tJ;                                 \
        if (lane == 0) DIr_w[k] = tI;                                                                                          \
    }

    if (64 * cA <= i) {
        const int kA1 = min(64 * cA + 64, i);                          // end of A's diagonal tile (exclusive), rows k < i only
        const int kB0 = haveB ? min(64 * cB, i) : i;                   // B joins here
        const int kB1 = haveB ? min(64 * cB + 64, i) : i;
        int k = 64 * cA;
        for (; k < kA1; ++k) JKP_ROW(JKP_DIAG, JKP_OFF)
        for (; k < kB0; ++k) JKP_ROW(JKP_FULL, JKP_OFF)
        for (; k < kB1; ++k) JKP_ROW(JKP_FULL, JKP_DIAG)
        for (; k < i; ++k) JKP_ROW(JKP_FULL, JKP_FULL)
        {   // k == i
            const long long bk = (long long)i * (i + 1) / 2;
            double pjk[JBB], rJ[JBB], rI = 0.0;
#pragma unroll
            for (int r = 0; r < JBB; ++r) { pjk[r] = sPj[r * N + i]; rJ[r] = 0.0; }
            const double pik = sPi[i];
            jkp_chunk_last(A, g, Tr, Pp, yg + bk, bk, pjk, pik, ppij, jd, rJ, rI);
            if (haveB && 64 * cB <= i) jkp_chunk_last(B, g, Tr, Pp, yg + bk, bk, pjk, pik, ppij, jd, rJ, rI);
            const double tJ = wave_sum8(rJ);
            const double tI = wave_sum1(rI);
            if ((lane & 7) == 0 && (lane >> 3) < g.nr) DJr_w[(size_t)(lane >> 3) * rowW + i] = tJ;
            if (lane == 0) DIr_w[i] = tI;
        }
    }
#undef JKP_ROW
    // column parts (every column l < N belongs to exactly one wave)
    if (A.l < N) {
        DIc[(size_t)blockIdx.x * N + A.l] = A.colI;
#pragma unroll
        for (int r = 0; r < JBB; ++r)
            if (r < g.nr) DJc[(size_t)(g.r0 + r) * N + A.l] = A.colJ[r];
    }
    if (haveB && B.l < N) {
        DIc[(size_t)blockIdx.x * N + B.l] = B.colI;
#pragma unroll
        for (int r = 0; r < JBB; ++r)
            if (r < g.nr) DJc[(size_t)(g.r0 + r) * N + B.l] = B.colJ[r];
    }
    {
        const double t = wave_sum8(jd);
        if ((lane & 7) == 0 && (lane >> 3) < g.nr) Jd[(size_t)(g.r0 + (lane >> 3)) * W + w] = t;
    }
}

// Jt partial sums: groups are sorted by descending partial length, so the groups that cover column q are a prefix of the list.
// grid (ceil(npair/256), SEG): segment s sums its slice of that prefix; out[s][q].
__global__ __launch_bounds__(256) void jt_reduce_kernel(const double *__restrict__ ypart, const JKGroup *__restrict__ groups, int n_groups,
                                                        long long npair, double *__restrict__ out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= npair) return;
    const int per = (n_groups + gridDim.y - 1) / gridDim.y;
    const int g0 = blockIdx.y * per, g1 = min(n_groups, g0 + per);
    double s = 0.0;
#pragma unroll 4
    for (int g = g0; g < g1; ++g) {
        if (groups[g].ylen <= q) break;
        s += ypart[groups[g].yoff + q];
    }
    out[(size_t)blockIdx.y * npair + q] = s;
}

// D[a][x] = sum over groups with i == a of (column part + row parts of the waves)[x] + the same over owned rows (i > a, j == a).
// grid (N, ceil(N/64)), 256 threads = 4 slices x 64 columns; gfirst[a]..gfirst[N+a] are the groups with i == a.
__global__ __launch_bounds__(256) void kd_reduce_kernel(const double *__restrict__ DIc, const double *__restrict__ DIr,
                                                        const double *__restrict__ DJc, const double *__restrict__ DJr, int W,
                                                        const int *__restrict__ gfirst, const int *__restrict__ rowmap, int N,
                                                        double *__restrict__ D)
{
    __shared__ double sPart[256];
    const int a = blockIdx.x;
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int x = blockIdx.y * 64 + lane;
    const int nw = min(W, (x >> 6) + 1);                  // waves whose first chunk starts at or before x
    double s = 0.0;
    if (x < N) {
        if (x <= a)
            for (int g = gfirst[a] + sl; g < gfirst[N + a]; g += 4) {
                double t = DIc[(size_t)g * N + x];
                for (int w = 0; w < nw; ++w) t += DIr[((size_t)g * W + w) * N + x];
                s += t;
            }
        for (int i = max(a + 1, x) + sl; i < N; i += 4) {
            const int r = rowmap[(size_t)i * (i + 1) / 2 + a];
            if (r < 0) continue;
            double t = DJc[(size_t)r * N + x];
            for (int w = 0; w < nw; ++w) t += DJr[((size_t)r * W + w) * N + x];
            s += t;
        }
    }
    sPart[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && x < N) D[(size_t)a * N + x] = ((sPart[lane] + sPart[64 + lane]) + sPart[128 + lane]) + sPart[192 + lane];
}

// K = D + D2^T (D2 = D for a symmetric density; for a general one D = D(P^T), D2 = D(P));
// J[i][j] = sum_w Jd[row(ij)][w] (owned rows) + sum_s Jt_s[pair(ij)]
__global__ void jk_packed_final_kernel(const double *__restrict__ D, const double *__restrict__ D2, const double *__restrict__ Jd, int W,
                                       const double *__restrict__ Jt, int nseg, const int *__restrict__ rowmap, int N,
                                       double *__restrict__ J, double *__restrict__ K)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int a = e / N, b = e - a * N;
    K[e] = D[e] + D2[(size_t)b * N + a];
    const int hi = max(a, b), lo = min(a, b);
    const long long p = (long long)hi * (hi + 1) / 2 + lo, npair = (long long)N * (N + 1) / 2;
    const int r = rowmap[p];
    double s = 0.0;
    if (r >= 0)
        for (int w = 0; w < W; ++w) s += Jd[(size_t)r * W + w];
    for (int t = 0; t < nseg; ++t) s += Jt[(size_t)t * npair + p];
    J[e] = s;
}

__device__ __forceinline__ void unpair(long long q, int &k, int &l)
{
    long long kk = (long long)((sqrt(8.0 * (double)q + 1.0) - 1.0) * 0.5);
    while (kk * (kk + 1) / 2 > q) --kk;
    while ((kk + 1) * (kk + 2) / 2 <= q) ++kk;
    k = (int)kk;
    l = (int)(q - kk * (kk + 1) / 2);
}

struct OutRowP {
    int i, j;              // output AO indices (i >= j)
    int cartA, cartB;      // first Cartesian AO of the two bra shells
    int ncb, pad;
    long long slab_off;    // first slab row of this bra pair
    long long dst_off;     // offset of the packed row in the stored tensor
};

// packed tensor row (i,j) = bra transform of the ket-transformed slab, keeping only pairs (k >= l) up to (i,j)
__global__ void xform_bra_store_packed(const double *__restrict__ in, double *__restrict__ eri, const OutRowP *__restrict__ rows,
                                       long long row_len, int ld, const int *__restrict__ ptr, const int *__restrict__ idx,
                                       const double *__restrict__ val)
{
    const OutRowP R = rows[blockIdx.y];
    const double *__restrict__ src = in + R.slab_off * row_len;
    double *__restrict__ dst = eri + R.dst_off;
    const long long len = (long long)R.i * (R.i + 1) / 2 + R.j + 1;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < len; q += (long long)gridDim.x * blockDim.x) {
        int k, l;
        unpair(q, k, l);
        const long long x = (long long)k * ld + l;
        double s = 0.0;
        for (int qa = ptr[R.i]; qa < ptr[R.i + 1]; ++qa) {
            const long long ra = (long long)(idx[qa] - R.cartA) * R.ncb;
            double t = 0.0;
            for (int qb = ptr[R.j]; qb < ptr[R.j + 1]; ++qb) t += val[qb] * src[(ra + (idx[qb] - R.cartB)) * row_len + x];
            s += val[qa] * t;
        }
        dst[q] = s;
    }
}

__device__ __forceinline__ double packed_element(const double *__restrict__ eri, const int *__restrict__ rowmap,
                                                 const long long *__restrict__ rowoff, int i, int j, int k, int l)
{
    const long long p = (long long)max(i, j) * (max(i, j) + 1) / 2 + min(i, j);
    const long long q = (long long)max(k, l) * (max(k, l) + 1) / 2 + min(k, l);
    const int r = rowmap[max(p, q)];
    return (r >= 0) ? eri[rowoff[r] + min(p, q)] : 0.0;
}

// packed -> dense N^4 with all images (what the reference leaves in ERI_AO, pyx:1335-1342).  On several ranks an element
// appears on the rank that owns row max(p,q); the others contribute zero (sum over ranks = dense tensor).
__global__ void expand_dense_packed_kernel(const double *__restrict__ eri, const int *__restrict__ rowmap, const long long *__restrict__ rowoff,
                                           int N, double *__restrict__ dense)
{
    const long long total = (long long)N * N * N * N;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(e % N);
        long long r = e / N;
        const int k = (int)(r % N); r /= N;
        const int j = (int)(r % N);
        const int i = (int)(r / N);
        dense[e] = packed_element(eri, rowmap, rowoff, i, j, k, l);
    }
}

__global__ void sample_packed_kernel(const double *__restrict__ eri, const int *__restrict__ rowmap, const long long *__restrict__ rowoff,
                                     long long n, const int *__restrict__ idx, double *__restrict__ out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    out[q] = packed_element(eri, rowmap, rowoff, idx[4 * q], idx[4 * q + 1], idx[4 * q + 2], idx[4 * q + 3]);
}

// full rows for the GEMM-shaped consumers (AO->MO): out[r - r0][k][l] (leading dimension ld) for local rows r0 <= r < r0 + nb
__global__ void unpack_full_rows_kernel(const double *__restrict__ eri, const int *__restrict__ rowmap, const long long *__restrict__ rowoff,
                                        const int2 *__restrict__ row_ij, long long r0, int nb, int N, int ld, double *__restrict__ out)
{
    const long long per = (long long)N * ld, total = (long long)nb * per;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long rr = e / per;
        const int rem = (int)(e - rr * per);
        const int k = rem / ld, l = rem - k * ld;
        const int2 ij = row_ij[r0 + rr];
        out[e] = (l < N) ? packed_element(eri, rowmap, rowoff, ij.x, ij.y, k, l) : 0.0;
    }
}
