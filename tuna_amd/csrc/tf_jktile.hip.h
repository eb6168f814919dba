// tf_jktile.hip.h -- Fock build from the "tiles" tensor layout (tf_tiles.h) on the FP64 matrix core.
// Reference: calculate_coulomb_matrix tuna_scf.py:55-72 ("ijkl,kl->ij"), calculate_exchange_matrix tuna_scf.py:27-44 ("ilkj,kl->ij").
// The reference keeps all 8 images of every (ij|kl) (pyx:1335-1342) and the zeros of the x/y parity rule (pyx:1324-1327); here each
// unique value is stored once and read once per build.
//
// One pass has to feed six outputs per stored element m = (ij|kl)  (i >= j, k >= l, (kl) <= (ij)):
//     Jd[ij] += m Pp[kl]        Jt[kl] += m Pp[ij]  (kl != ij)                            Pp[xy] = P[x][y] + P[y][x], or P[x][x]
//     D[i][k] += m P[j][l]      D[i][l] += m P[j][k] (k != l)      D[j][k] += m P[i][l] (i != j)      D[j][l] += m P[i][k] (i != j, k != l)
// (the D terms at half weight when kl == ij);  J = Jd + Jt,  K = D + D^T for a symmetric P.
//
// jk_tile_kernel: a workgroup = one task (first index i, class pair (a, b), strip of <= 64 rows k, <= 4 column blocks of 16 l); wave w
// owns column block lb0 + w; the loop runs over the rows' second index j -- one contiguous slice of the task's region per step.  A wave
// holds its piece of the slice in the A-operand layout of v_mfma_f64_16x16x4 (lane = (row k = lane & 15, lane group kk = lane >> 4),
// four registers = columns l = 4 kk + r): MB row blocks of 16 x 16 values.
//   * contraction over l -- D[j][k] = sum_l m P[i][l] -- on the matrix core: B operand = P_d[i][l] in column d (one MFMA pass serves
//     every density of the pass); the 4 waves' results are merged in LDS every TT_KB steps and written once per (row, strip, chunk);
//   * everything indexed by l -- Jt[kl], D[i][l], D[j][l] -- and the sums that run over j are lane-local multiply-adds; what has to
//     cross lanes crosses the 16 lanes of a DPP row once per step (D[j][l]: four values, a transposing butterfly) or once per task;
//   * Jd[ij]: lane-local products, four steps summed over the wave together.
// No atomics: every partial sum has one owner and the reductions (jk_tile_reduce_kernel) add in fixed order -- bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include "tf_layout.hip.h"
#include "tf_tiles.h"
#include "tf_jkpacked.hip.h"      // buffer loads / stores, DPP and permlane sums

typedef double tt_v4d __attribute__((ext_vector_type(4)));

// (ij|kl) from the stored tensor, ORIGINAL indices; rows owned by another rank read as 0 (sum over ranks = the tensor).
__device__ __forceinline__ double tile_element(const double *__restrict__ eri, const TView &V, const BLayout &L, int i, int j, int k, int l)
{
    int ih = max(i, j), il = min(i, j), kh = max(k, l), kl = min(k, l);
    const int wi = L.ao[ih], wj = L.ao[il], wk = L.ao[kh], wl = L.ao[kl];
    if ((ao_cls(wi) ^ ao_cls(wj)) != (ao_cls(wk) ^ ao_cls(wl))) return 0.0;     // x/y parity, pyx:1324-1327
    const long long p = (long long)ih * (ih + 1) / 2 + il, q = (long long)kh * (kh + 1) / 2 + kl;
    int r1 = wi, r2 = wj, c1 = wk, c2 = wl;
    if (q > p) { r1 = wk; r2 = wl; c1 = wi; c2 = wj; }
    const long long ad = tt_elem_addr(V, L.clsI, ao_sigma(L, r1), ao_sigma(L, r2), ao_sigma(L, c1), ao_sigma(L, c2));
    return ad < 0 ? 0.0 : eri[ad];
}

__global__ void expand_dense_tiles_kernel(const double *__restrict__ eri, TView V, BLayout L, double *__restrict__ dense)
{
    const int N = L.N;
    const long long total = (long long)N * N * N * N;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(e % N);
        long long r = e / N;
        const int k = (int)(r % N); r /= N;
        dense[e] = tile_element(eri, V, L, (int)(r / N), (int)(r % N), k, l);
    }
}

__global__ void sample_tiles_kernel(const double *__restrict__ eri, TView V, BLayout L, long long n, const int *__restrict__ idx, double *__restrict__ out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    out[q] = tile_element(eri, V, L, idx[4 * q], idx[4 * q + 1], idx[4 * q + 2], idx[4 * q + 3]);
}

// ---- writing the tensor -------------------------------------------------------------------------------------------------------
// Bra transform of the generation slab (rows in the complete-row shape of their class: tf_jkpacked.hip.h, xform_bra_store_packed) into
// the tiles layout: thread x = a pair (k >= l) of the row's class; the element goes to its slot of the (i, pair, strip, chunk) region
// or to the edge triangle.  The tensor was zeroed before (pad slots).
struct OutRowT {
    int i, j;              // output AO indices (i >= j), original order: rows of the Cartesian -> spherical CSR
    int iI, jI;            // internal
    int c, ncb;            // class of the row; components of the second bra shell
    int cartA, cartB;      // first Cartesian AO of the two bra shells
    long long slab_off;    // first slab row of this bra pair
};

__global__ __launch_bounds__(256) void xform_bra_store_tiles(const double *__restrict__ in, double *__restrict__ eri,
                                                             const OutRowT *__restrict__ rows, long long RLS, BLayout L, TView V,
                                                             const int *__restrict__ ptr, const int *__restrict__ idx,
                                                             const double *__restrict__ val)
{
    __shared__ double sValA[32], sValB[32];
    __shared__ long long sOffA[32], sOffB[32];
    const OutRowT R = rows[blockIdx.y];
    const int c = R.c;
    if ((long long)blockIdx.x * 256 >= bl_np(L, c)) return;
    const int pa = ptr[R.i], na = min(32, ptr[R.i + 1] - pa), pb = ptr[R.j], nb = min(32, ptr[R.j + 1] - pb);
    if (threadIdx.x < na) { sValA[threadIdx.x] = val[pa + threadIdx.x]; sOffA[threadIdx.x] = (long long)(idx[pa + threadIdx.x] - R.cartA) * R.ncb * RLS; }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + nb) {
        const int t = threadIdx.x - 64;
        sValB[t] = val[pb + t]; sOffB[t] = (long long)(idx[pb + t] - R.cartB) * RLS;
    }
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x;                            // pair index inside a complete class-c row
    if (x >= bl_np(L, c)) return;
    const int kI = L.gk[bl_gbase(L, c) + x / TF_SEG_PAD];
    const int a = L.clsI[kI];
    const KInfo ki = L.kinfo[(size_t)c * L.N + kI];
    const int lam = x - bl_fullsec(L, c, a) - ki.offA;
    const int lamj = R.jI - bl_cstart(L, L.clsI[R.jI]);
    if (lam >= ki.cnt || kI - bl_cstart(L, a) >= L.cntA[(size_t)a * L.N + R.iI] || (kI == R.iI && lam > lamj)) return;   // padding; k > i; (kl) > (ij)
    const long long ad = tt_elem_addr(V, L.clsI, R.iI, R.jI, kI, bl_cstart(L, a ^ c) + lam);
    if (ad < 0) return;
    const double *__restrict__ src = in + R.slab_off * RLS + x;
    double s = 0.0;
    for (int qa = 0; qa < na; ++qa) {
        double t = 0.0;
        for (int qb = 0; qb < nb; ++qb) t += sValB[qb] * src[sOffA[qa] + sOffB[qb]];
        s += sValA[qa] * t;
    }
    eri[ad] = s;
}

// The stored part of local rows as symmetric matrices, for the GEMM-shaped consumers (AO->MO): the tiles counterparts of
// unpack_own_rows_kernel / unpack_own_rows_blocked_kernel (tf_jkpacked.hip.h).
__global__ void unpack_own_rows_tiles_kernel(const double *__restrict__ eri, TView V, BLayout L, const int2 *__restrict__ row_ij, long long r0, int ld,
                                             double *__restrict__ out)
{
    const long long r = r0 + blockIdx.y;
    const int2 ij = row_ij[r];
    const int wi = L.ao[ij.x], wj = L.ao[ij.y];
    const int c = ao_cls(wi) ^ ao_cls(wj), iI = ao_sigma(L, wi), jI = ao_sigma(L, wj), lamj = ao_loc(wj);
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= bl_np(L, c)) return;
    const int kI = L.gk[bl_gbase(L, c) + x / TF_SEG_PAD];
    const int a = L.clsI[kI];
    const KInfo ki = L.kinfo[(size_t)c * L.N + kI];
    const int lam = x - bl_fullsec(L, c, a) - ki.offA;
    if (lam >= ki.cnt || kI - bl_cstart(L, a) >= L.cntA[(size_t)a * L.N + iI] || (kI == iI && lam > lamj)) return;
    const int lI = bl_cstart(L, a ^ c) + lam;
    const long long ad = tt_elem_addr(V, L.clsI, iI, jI, kI, lI);
    if (ad < 0) return;
    double v = eri[ad];
    if (kI == iI && lam == lamj) v *= 0.5;
    const int k = L.origI[kI], l = L.origI[lI];
    double *__restrict__ o = out + (size_t)blockIdx.y * L.N * ld;
    o[(size_t)k * ld + l] = v;
    o[(size_t)l * ld + k] = v;
}

__global__ void unpack_own_rows_blocked_tiles_kernel(const double *__restrict__ eri, TView V, BLayout L, const int2 *__restrict__ row_ij,
                                                     const int *__restrict__ rows, int c, RowBlocks RBk, double *__restrict__ out)
{
    const long long r = rows[blockIdx.y];
    const int2 ij = row_ij[r];
    const int wi = L.ao[ij.x], wj = L.ao[ij.y];
    const int iI = ao_sigma(L, wi), jI = ao_sigma(L, wj), lamj = ao_loc(wj);
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= bl_np(L, c)) return;
    const int kI = L.gk[bl_gbase(L, c) + x / TF_SEG_PAD];
    const int a = L.clsI[kI], b = a ^ c;
    const KInfo ki = L.kinfo[(size_t)c * L.N + kI];
    const int lam = x - bl_fullsec(L, c, a) - ki.offA;
    const int kl = kI - bl_cstart(L, a);
    if (lam >= ki.cnt || kl >= L.cntA[(size_t)a * L.N + iI] || (kI == iI && lam > lamj)) return;
    const long long ad = tt_elem_addr(V, L.clsI, iI, jI, kI, bl_cstart(L, b) + lam);
    if (ad < 0) return;
    double v = eri[ad];
    if (kI == iI && lam == lamj) v *= 0.5;
    double *__restrict__ o = out + (size_t)blockIdx.y * RBk.rstride;
    o[RBk.boff[a] + (size_t)kl * RBk.ldb[a] + lam] = v;
    o[RBk.boff[b] + (size_t)lam * RBk.ldb[b] + kl] = v;
}

// ---- densities ----------------------------------------------------------------------------------------------------------------
// X[sigma(r)][sigma(c)] = P[r][c] (or P[c][r]); pair matrices Pm: block of class pair p = [k loc][pitch] with Pp(k, l) = P[k][l] + P[l][k]
// (k != l) or P[k][k]; pad columns zero (set once at allocation).
__global__ void pack_density_tiles_kernel(const double *__restrict__ P, BLayout L, TView V, int transpose, double *__restrict__ X, double *__restrict__ Pm)
{
    const int N = L.N;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int k = e / N, l = e - k * N;
    const double a = P[e], b = P[(size_t)l * N + k];
    const int wk = L.ao[k], wl = L.ao[l];
    X[(size_t)ao_sigma(L, wk) * N + ao_sigma(L, wl)] = transpose ? b : a;
    const int p = V.tab[TVT_PID + ao_cls(wk) * 4 + ao_cls(wl)];
    if (ao_cls(wk) == V.tab[TVT_PA + p]) Pm[V.tab[TVT_PMOFF + p] + ao_loc(wk) * V.tab[TVT_PMPITCH + p] + ao_loc(wl)] = (k == l) ? a : a + b;   // (a triangle pair: both orders)
}

// ---- the Fock kernel ------------------------------------------------------------------------------------------------------------
struct TJArgs {
    double *DJ, *Jt, *Jd, *DIk, *DIl;
    size_t sDJ, sJt, sJd, sDIk, sDIl;   // strides between the densities of a pass
    int N, pm_len;
};

// four per-lane values -> the sums over the 16 lanes of every DPP row; lane m of a row holds the total of value m >> 2
__device__ __forceinline__ double row_sum4(double v0, double v1, double v2, double v3)
{
    const double u0 = pair_step8(v0, v2), u1 = pair_step8(v1, v3);     // lanes 0-7: v0 / v1; lanes 8-15: v2 / v3
    return quad_sum(pair_step4(u0, u1));                                 // bit 2 clear: u0; set: u1
}
// four per-lane values -> lane L holds the wave total of value L >> 4
__device__ __forceinline__ double wave_sum4(double v0, double v1, double v2, double v3)
{
    const double w0 = pair_step32(v0, v2), w1 = pair_step32(v1, v3);   // lanes 0-31: v0 / v1; 32-63: v2 / v3
    double t = pair_step16(w0, w1);                                      // row q of 16 lanes: value q
    t += dpp_merge<TF_DPP_ROR8, 0xF>(t, t);
    return sum8(t);
}

// LDS of a workgroup: the row-sum slots of TT_KB steps, the Jd weights of every wave's tile, and the density rows of the next TT_KB steps
// (P_d[j][k] of the strip, P_d[j][l] of the chunk, P_d[i][j], P_d[j][i]: staged by all waves together -- vector memory operations
// complete in order, so nothing a step needs may be loaded at the step itself: it would wait for the prefetched slice).
#define TT_PST (TT_KS + TT_LB + 4)                // doubles of one staged row of a wave: 64 k + 16 l + {P[i][j], P[j][i]} (+ pad)
template <int ND, int MB>
struct TJLds {
    double slots[TT_KB * TT_W * ND * 64];
    double2 pp[TT_W][ND][MB * 2][64];
    double pst[TT_W][ND][TT_KB][TT_PST];         // every wave stages its own rows: no barrier between the staging and its use
};

template <int ND, int MB, bool DIAG>
__device__ __forceinline__ void tj_run(const double *__restrict__ T, const double *__restrict__ X, const double *__restrict__ Pm, const TJArgs &A,
                                       const TTask &t, int w, int lane, bool active, TJLds<ND, MB> &S)
{
    const int N = A.N;
    const int m = lane & 15, kk = lane >> 4;
    const bool tri = t.a == t.b;
    const int lb = t.lb0 + w, ks = t.k0 / TT_KS;
    const int iI = t.i, nj = t.nj, nwg = (int)(blockDim.x >> 6);
    const size_t nn = (size_t)N * N;
    // ---- per-lane constants
    unsigned offA[MB], offB[MB];
    double pik[ND][MB];
    int rdiag[MB];
    bool rv[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int r = 16 * mb + m;
        rv[mb] = active && r < t.nks;
        const int rs = t.roff0 + r;
        const int rl = rv[mb] ? tt_row_len(tri, ks, lb, rs, t.nl) : 0;
        const unsigned o = 8u * (unsigned)(t.woff[w] + tt_row_off(tri, ks, lb, rs, t.nl) + 4 * kk);
        offA[mb] = (4 * kk < rl) ? o : TF_BUF_OOB;
        offB[mb] = (4 * kk + 2 < rl) ? o + 16u : TF_BUF_OOB;
        const int kI = t.kbase + t.k0 + r;
        rdiag[mb] = (t.k0 + r) - (TT_LB * lb + 4 * kk);                  // the register of this lane that holds k == l (triangles)
#pragma unroll
        for (int d = 0; d < ND; ++d) pik[d][mb] = rv[mb] ? X[d * nn + (size_t)iI * N + kI] : 0.0;
    }
    const int lcol0 = TT_LB * lb + 4 * kk;                                 // loc of the lane's first column
    const int ncol = t.ncol;
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(T + t.base), 0, (int)min((long long)nj * t.slice * 8, 0x7fffffffLL), 0x00020000);
    // B operand of the row sums: column d = P_d[i][l] (lanes m == d), zero elsewhere
    double b1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[r] = (active && m < ND && lcol0 + r < ncol) ? X[(size_t)(m < ND ? m : 0) * nn + (size_t)iI * N + t.lbase + lcol0 + r] : 0.0;
    // weights of Jd: Pp_d(k, l) of the lane's elements, kept in LDS (pad columns of the pair matrices are zero)
    if (active) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const double *src = Pm + (size_t)d * A.pm_len + t.pm_off + (size_t)(t.k0 + 16 * mb + m) * t.pm_pitch + lcol0;
                S.pp[w][d][2 * mb][lane] = (rv[mb] && lcol0 < t.pm_pitch) ? *reinterpret_cast<const double2 *>(src) : make_double2(0.0, 0.0);
                S.pp[w][d][2 * mb + 1][lane] = (rv[mb] && lcol0 + 2 < t.pm_pitch) ? *reinterpret_cast<const double2 *>(src + 2) : make_double2(0.0, 0.0);
            }
    }
    // accumulators
    double jt[ND][MB][4], x5[ND][4], x2[ND][MB], jdb[ND][TT_KB];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x5[d][r] = 0.0;
#pragma unroll
        for (int q = 0; q < TT_KB; ++q) jdb[d][q] = 0.0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            x2[d][mb] = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) jt[d][mb][r] = 0.0;
        }
    }
    // staging of the density rows of the steps sb .. sb + TT_KB - 1 (this wave's copy): lane = row k of the strip for each of the steps;
    // lane = (step, column) for the wave's 16 columns; lanes 0-7 = (step, P[i][j] / P[j][i])
    double stk[ND][TT_KB], stl[ND], stx[ND];
    auto stage_load = [&](int sb) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double *Xd = X + d * nn;
#pragma unroll
            for (int kq = 0; kq < TT_KB; ++kq) {
                const int s = sb + kq;
                stk[d][kq] = (active && s < nj && lane < t.nks) ? Xd[(size_t)(t.j0 + s) * N + t.kbase + t.k0 + lane] : 0.0;
            }
            const int sq = sb + (lane >> 4), lc = TT_LB * lb + (lane & 15);
            stl[d] = (active && sq < nj && lc < ncol) ? Xd[(size_t)(t.j0 + sq) * N + t.lbase + lc] : 0.0;
            const int sx = sb + (lane >> 1);
            const bool okx = active && lane < 2 * TT_KB && sx < nj;
            stx[d] = !okx ? 0.0 : ((lane & 1) ? Xd[(size_t)(t.j0 + sx) * N + iI] : Xd[(size_t)iI * N + t.j0 + sx]);
        }
    };
    auto stage_store = [&]() {
        if (!active) return;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
#pragma unroll
            for (int kq = 0; kq < TT_KB; ++kq) S.pst[w][d][kq][lane] = stk[d][kq];
            S.pst[w][d][lane >> 4][TT_KS + (lane & 15)] = stl[d];
            if (lane < 2 * TT_KB) S.pst[w][d][lane >> 1][TT_KS + TT_LB + (lane & 1)] = stx[d];
        }
    };
    // the wave's piece of a slice lives in ONE set of registers: the loads of row block mb of the next step are issued as soon as
    // row block mb of this step has been consumed (a full step of distance, nothing in flight twice)
    auto load_mb = [&](double2 (&B)[MB][2], int mb, int s) {
        const unsigned so = (unsigned)s * (unsigned)t.slice * 8u;
        B[mb][0] = buf_load2<2>(rt, offA[mb], so);
        B[mb][1] = buf_load2<2>(rt, offB[mb], so);
    };
    auto step = [&](double2 (&B)[MB][2], int s, int kq) {
        const bool self = t.self_last && s == nj - 1;
        double ppij[ND], pjl[ND][4], pjk[ND][MB];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double *ps = S.pst[w][d][kq];
            const double xij = ps[TT_KS + TT_LB], xji = ps[TT_KS + TT_LB + 1];
            ppij[d] = self ? xij : xij + xji;
            const double2 q0 = *reinterpret_cast<const double2 *>(ps + TT_KS + 4 * kk), q1 = *reinterpret_cast<const double2 *>(ps + TT_KS + 4 * kk + 2);
            pjl[d][0] = q0.x; pjl[d][1] = q0.y; pjl[d][2] = q1.x; pjl[d][3] = q1.y;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) pjk[d][mb] = ps[16 * mb + m];
        }
        double t4[ND][4];
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) t4[d][r] = 0.0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const double tv[4] = {B[mb][0].x, B[mb][0].y, B[mb][1].x, B[mb][1].y};
            load_mb(B, mb, min(s + 1, nj - 1));
            tt_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[r], b1[r], acc, 0, 0, 0);
            // row sums D_d[j][k]: column d of the result tile = lanes m == d, rows kk + 4 q
            if (m < ND) {
                double *sl = S.slots + ((size_t)(kq * TT_W + w) * ND + m) * 64 + mb * 16 + kk * 4;
                *reinterpret_cast<double2 *>(sl) = make_double2(acc[0], acc[1]);
                *reinterpret_cast<double2 *>(sl + 2) = make_double2(acc[2], acc[3]);
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const double2 w0 = S.pp[w][d][2 * mb][lane], w1 = S.pp[w][d][2 * mb + 1][lane];
                const double ppv[4] = {w0.x, w0.y, w1.x, w1.y};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double tm = tv[r];
                    const double to = (DIAG && r == rdiag[mb]) ? 0.0 : tm;          // without the diagonal k == l
                    jt[d][mb][r] += tm * ppij[d];
                    jdb[d][kq] += tm * ppv[r];
                    x2[d][mb] += tm * pjl[d][r];
                    x5[d][r] += to * pjk[d][mb];
                    t4[d][r] += to * pik[d][mb];
                }
            }
        }
        // column sums D_d[j][l]: over the 16 rows of the lane's DPP row; lane m then holds column 4 kk + (m >> 2)
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double cs = row_sum4(t4[d][0], t4[d][1], t4[d][2], t4[d][3]);
            if ((m & 3) == 0 && !self) A.DJ[d * A.sDJ + t.dj_base + (long long)s * t.dj_len + t.dj_loff[w] + 4 * kk + (m >> 2)] = cs;
        }
    };
    double2 R0[MB][2];
    stage_load(0);
    if (active) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) load_mb(R0, mb, 0);
    }
    stage_store();
    __syncthreads();                                                       // (also: the Jd weights are in place)
    for (int sb = 0; sb < nj; sb += TT_KB) {
        stage_load(sb + TT_KB);                                            // the density rows of the next block (written behind the first barrier)
        if (active) {
#pragma unroll
            for (int kq = 0; kq < TT_KB; ++kq) {
                const int s = sb + kq;
                if (s < nj) step(R0, s, kq);
            }
            // Jd of the steps sb .. sb + 3: the wave totals of four values together
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const double v = wave_sum4(jdb[d][0], jdb[d][1], jdb[d][2], jdb[d][3]);
                if ((lane & 15) == 0 && sb + (lane >> 4) < nj) A.Jd[d * A.sJd + t.jd_base + (size_t)w * nj + sb + (lane >> 4)] = v;
#pragma unroll
                for (int q = 0; q < TT_KB; ++q) jdb[d][q] = 0.0;
            }
        }
        jkp_lds_barrier();
        // merge of the row sums: wave w adds the waves' partials of step sb + w (+ nwg ..) and writes the K slot of that row
        for (int kq = w; kq < TT_KB; kq += nwg) {
            const int s = sb + kq;
            if (s >= nj) break;
            const bool self = t.self_last && s == nj - 1;
            const int krel = 16 * (lane >> 4) + ((lane >> 2) & 3) + 4 * (lane & 3);
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                double v = S.slots[((size_t)(kq * TT_W) * ND + d) * 64 + lane];
                for (int u = 1; u < t.nw; ++u) v += S.slots[((size_t)(kq * TT_W + u) * ND + d) * 64 + lane];
                if (krel < t.nks && (lane >> 4) < MB && !self) A.DJ[d * A.sDJ + t.dj_base + (long long)s * t.dj_len + t.dj_koff + krel] = v;
            }
        }
        stage_store();
        jkp_lds_barrier();
    }
    if (!active) return;
    // ---- what was summed over the steps
#pragma unroll
    for (int d = 0; d < ND; ++d) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {                                   // the Jt tile
            double *dst = A.Jt + d * A.sJt + t.jt_base + (long long)(t.k0 + 16 * mb + m) * t.jt_pitch + lcol0;
            if (rv[mb] && lcol0 < t.jt_pitch) *reinterpret_cast<double2 *>(dst) = make_double2(jt[d][mb][0], jt[d][mb][1]);
            if (rv[mb] && lcol0 + 2 < t.jt_pitch) *reinterpret_cast<double2 *>(dst + 2) = make_double2(jt[d][mb][2], jt[d][mb][3]);
        }
        const double cs = row_sum4(x5[d][0], x5[d][1], x5[d][2], x5[d][3]);   // D[i][l]: over the rows
        if ((m & 3) == 0) A.DIl[d * A.sDIl + (size_t)(t.di_base + w) * 16 + 4 * kk + (m >> 2)] = cs;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {                                   // D[i][k]: over the four lane groups
            double v = x2[d][mb];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (kk == 0) A.DIk[d * A.sDIk + (size_t)(t.di_base + w) * 64 + 16 * mb + m] = v;
        }
    }
}

// One workgroup per task; blockDim = 64 x (waves of the launch's bucket: >= the task's column blocks).  MB = row blocks of 16 of a
// task's strip (4: strips of 64 rows, one density; fewer for the passes over several densities: tf_tiles.h, `ksub`).
template <int ND, int MB>
__global__ __launch_bounds__(64 * TT_W, (MB >= 4 ? 1 : 2)) void jk_tile_kernel(const double *__restrict__ T, const TTask *__restrict__ tasks,
                                                               const double *__restrict__ X, const double *__restrict__ Pm, TJArgs A)
{
    __shared__ TJLds<ND, MB> S;
    const TTask &t = tasks[blockIdx.x];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool active = w < t.nw;
    // a triangle's tile holds diagonal elements k == l only where its rows and columns overlap
    const int lb = t.lb0 + w;
    const bool diag = t.a == t.b && active && TT_LB * lb <= t.k0 + t.nks - 1 && TT_LB * lb + TT_LB - 1 >= t.k0;
    if (diag) tj_run<ND, MB, true>(T, X, Pm, A, t, w, lane, active, S);
    else tj_run<ND, MB, false>(T, X, Pm, A, t, w, lane, active, S);
}

// ---- the edge elements m = (ij|il), l <= j (k == i): one workgroup per first index i ---------------------------------------------------
// Per-i outputs, internal indices: EJ[i][x] (J of the pair (i, x)), ED[i][x] (D[i][x]), EDT[i][x] (D[x][i]).  The term D[j][l] += w m P[i][i]
// runs over i and is added by the D[j][.] blocks of jk_tile_reduce_kernel.  Fixed summation order per output.
struct TEArgs {
    const TRunI *runs;            // [N][4]
    long long edge_base;
    int N;
    const int *tab;               // TView::tab
    double *EJ, *ED, *EDT;
    size_t sE;                    // stride between densities
};
template <int ND>
__global__ __launch_bounds__(256) void jk_edge_kernel(const double *__restrict__ T, const double *__restrict__ X, const int *__restrict__ clsI, TEArgs A)
{
    __shared__ double sred[256];
    const int N = A.N, iI = blockIdx.x;
    const size_t nn = (size_t)N * N;
    const int ci = clsI[iI];
    for (int d = 0; d < ND; ++d) {
        const double *Xd = X + d * nn;
        double dii = 0.0;                                                  // this thread's share of D[i][i]
        for (int x = threadIdx.x; x < N; x += 256) {
            const int cx = clsI[x];
            const int c0 = A.tab[TVT_CSTART + cx];
            const TRunI R = A.runs[(size_t)iI * 4 + cx];
            const int lx = x - c0, l0 = R.j0 - c0;
            double ej = 0.0, ed = 0.0, edt = 0.0;
            if (R.nj > 0) {
                const double *E = T + A.edge_base + R.e_base - (long long)l0 * (l0 + 1) / 2;     // E[T(loc j) + loc l]
                const double ppix = (x == iI) ? Xd[(size_t)iI * N + iI] : Xd[(size_t)iI * N + x] + Xd[(size_t)x * N + iI];
                // x as j: the row (i, x), all l <= x
                if (lx >= l0 && lx < l0 + R.nj) {
                    const double *Ej = E + (long long)lx * (lx + 1) / 2;
                    for (int ll = 0; ll <= lx; ++ll) {
                        const int lI = c0 + ll;
                        const double mv = Ej[ll], wgt = (ll == lx) ? 0.5 : 1.0;
                        const double ppil = (lI == iI) ? Xd[(size_t)iI * N + iI] : Xd[(size_t)iI * N + lI] + Xd[(size_t)lI * N + iI];
                        ej += mv * ppil;                                   // Jd[ij] += m Pp[il]
                        dii += wgt * mv * Xd[(size_t)x * N + lI];          // D[i][i] += w m P[j][l]
                        if (x != iI) edt += wgt * mv * Xd[(size_t)iI * N + lI];   // D[j][i] += w m P[i][l]   (i != j)
                    }
                }
                // x as l: the rows (i, j), j >= x of x's class
                for (int lj = max(lx, l0); lj < l0 + R.nj; ++lj) {
                    const int jI = c0 + lj;
                    const double mv = E[(long long)lj * (lj + 1) / 2 + lx], wgt = (lj == lx) ? 0.5 : 1.0;
                    if (lj != lx) {
                        const double ppij = (jI == iI) ? Xd[(size_t)iI * N + iI] : Xd[(size_t)iI * N + jI] + Xd[(size_t)jI * N + iI];
                        ej += mv * ppij;                                   // Jt[il] += m Pp[ij]   (l != j)
                    }
                    if (x != iI) ed += wgt * mv * Xd[(size_t)jI * N + iI]; // D[i][l] += w m P[j][i]   (l != i)
                }
                (void)ppix;
            }
            A.EJ[d * A.sE + (size_t)iI * N + x] = ej;
            A.ED[d * A.sE + (size_t)iI * N + x] = ed;
            A.EDT[d * A.sE + (size_t)iI * N + x] = edt;
        }
        sred[threadIdx.x] = dii;
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = 0.0;
            for (int q = 0; q < 256; ++q) s += sred[q];
            A.ED[d * A.sE + (size_t)iI * N + iI] += s;
        }
        __syncthreads();
    }
    (void)ci;
}

// ---- reductions: one launch, three kinds of blocks, fixed summation order inside every block ------------------------------------
struct TRArgs {
    const TTask *tasks;           // launch order (the task list of the pass)
    const TPairI *pairs;          // [N][10] of the list
    const TRunI *runs;            // [N][4]
    const int *itask_ptr, *itasks, *jlist_ptr, *jlist;
    const int *clsI, *origI;
    const int *cntA;              // [4][N]
    const double *DJ, *Jt, *Jd, *DIk, *DIl, *T, *X;
    size_t sDJ, sJt, sJd, sDIk, sDIl, sO;
    double *Dj, *Di, *JtTot, *JD; // Dj[j][x], Di[i][x], JD[i][j]: [N][N] internal; JtTot: pair matrices
    long long edge_base;
    int N, ksub, npair, nd, pm_len;
    const int *tab;               // TView::tab
    int jt_rows;                  // rows of all pair matrices together (blocks of kind 1)
    int xtiles;                   // 64-column tiles of a row of N
};
#define TT_RED_THREADS 256

// kind 0: D[j][x] for one j and 64 columns x of one class: the DJ vectors of the rows (i, j), i != j (jlist) + the edge term
__device__ __forceinline__ void tr_dj_block(const TRArgs &R, int d, int jI, int X0, double *sPart)
{
    const int N = R.N, lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int x = X0 + lane;
    const int cX = R.clsI[X0], cj = R.clsI[jI];
    const int xs = R.tab[TVT_CSTART + cX], nX = R.tab[TVT_CSIZE + cX];
    const bool in = x < xs + nX;
    const int lx = x - xs, lj = jI - R.tab[TVT_CSTART + cj];
    const double *DJ = R.DJ + d * R.sDJ;
    const double *Xd = R.X + (size_t)d * N * N;
    double acc = 0.0;
    for (int q = R.jlist_ptr[jI] + sl; q < R.jlist_ptr[jI + 1]; q += TT_RED_THREADS / 64) {
        const int iI = R.jlist[q];
        const int c = R.clsI[iI] ^ cj;
        const int p = R.tab[TVT_PID + cX * 4 + (cX ^ c)];
        const TPairI P = R.pairs[(size_t)iI * 10 + p];
        const TRunI Rn = R.runs[(size_t)iI * 4 + cj];
        double v = 0.0;
        if (P.first_task >= 0 && in) {
            const double *vec = DJ + Rn.dj_base + (long long)(jI - Rn.j0) * Rn.dj_len;
            const int a = R.tab[TVT_PA + p], b = R.tab[TVT_PB + p];
            const bool tri = a == b;
            if (cX == a && lx < P.nk) {                                   // x as a row index k: the chunks of its stored strip
                const int ks = lx / TT_KS, nst = min(TT_KS, P.nk - TT_KS * ks);
                int nch, w;
                tt_chunks(tt_nlb(tri, ks, P.nk, P.nl), &nch, &w);
                const double *src = vec + P.dj_k + tt_dj_koff(tri, ks, P.nl) + (lx - TT_KS * ks);
                for (int ch = 0; ch < nch; ++ch) v += src[ch * nst];
            }
            if (cX == b && lx < P.nl) {                                   // x as a column index l: the sub-strips that reach its block
                const int lb = lx / TT_LB, ns = (P.nk + R.ksub - 1) / R.ksub, f = tt_dj_first_sub(tri, lb, R.ksub);
                const double *src = vec + P.dj_l + tt_dj_loff(tri, lb, P.nk, R.ksub) + (lx - TT_LB * lb);
                for (int s2 = f; s2 < ns; ++s2) v += src[(s2 - f) * TT_LB];
            }
        }
        if (in && cX == cj && lx <= lj) {                                   // edge: D[j][l] += w (ij|il) P[i][i]
            const long long l0 = Rn.j0 - R.tab[TVT_CSTART + cj];
            const double mv = R.T[R.edge_base + Rn.e_base + (long long)lj * (lj + 1) / 2 - l0 * (l0 + 1) / 2 + lx];
            v += (lx == lj ? 0.5 : 1.0) * mv * Xd[(size_t)iI * N + iI];
        }
        acc += v;
    }
    sPart[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && in) {
        double a2 = sPart[lane];
        for (int q = 1; q < TT_RED_THREADS / 64; ++q) a2 += sPart[64 * q + lane];
        R.Dj[d * R.sO + (size_t)jI * N + x] = a2;
    }
}

// kind 1: Jt totals of one row k of a class pair, 64 columns: the partial blocks of every first index above (k, l)
__device__ __forceinline__ void tr_jt_block(const TRArgs &R, int d, int rowid, int tile, double *sPart)
{
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    int p = 0, row = rowid;
    while (p + 1 < R.npair && row >= R.tab[TVT_CSIZE + R.tab[TVT_PA + p]]) { row -= R.tab[TVT_CSIZE + R.tab[TVT_PA + p]]; ++p; }     // (wave-uniform walk over <= 10 pairs)
    const int a = R.tab[TVT_PA + p], b = R.tab[TVT_PB + p];
    const bool tri = a == b;
    const int lc = 64 * tile + lane;
    const int nb = R.tab[TVT_CSIZE + b];
    if (64 * tile >= nb) return;
    const double *Jt = R.Jt + d * R.sJt;
    double acc = 0.0;
    for (int iI = sl; iI < R.N; iI += TT_RED_THREADS / 64) {
        const TPairI P = R.pairs[(size_t)iI * 10 + p];
        if (P.first_task < 0 || row >= P.nk) continue;
        if (lc < (tri ? row + 1 : P.nl)) {
            const double *src = Jt + P.jt_base + (long long)row * P.jt_pitch + lc;
            for (int part = 0; part < P.nparts; ++part) acc += src[part * P.jt_part_stride];
        }
    }
    sPart[threadIdx.x] = acc;
    __syncthreads();
    if (sl == 0 && lc < R.tab[TVT_PMPITCH + p]) {
        double a2 = sPart[lane];
        for (int q = 1; q < TT_RED_THREADS / 64; ++q) a2 += sPart[64 * q + lane];
        R.JtTot[(size_t)d * R.pm_len + R.tab[TVT_PMOFF + p] + (size_t)row * R.tab[TVT_PMPITCH + p] + lc] = a2;
    }
}

// kind 2: D[i][x] and Jd[i][j] for one i: the per-task outputs of the tasks of i (creation order)
__device__ __forceinline__ void tr_gather_block(const TRArgs &R, int d, int iI)
{
    const int N = R.N;
    for (int x = threadIdx.x; x < N; x += TT_RED_THREADS) {
        const int cX = R.clsI[x], lx = x - R.tab[TVT_CSTART + cX];
        double di = 0.0, jd = 0.0;
        for (int q = R.itask_ptr[iI]; q < R.itask_ptr[iI + 1]; ++q) {
            const TTask &t = R.tasks[R.itasks[q]];
            if (cX == t.a && lx >= t.k0 && lx < t.k0 + t.nks)
                for (int w = 0; w < t.nw; ++w) di += R.DIk[d * R.sDIk + (size_t)(t.di_base + w) * 64 + (lx - t.k0)];
            if (cX == t.b) {
                const int w = lx / TT_LB - t.lb0;
                if (w >= 0 && w < t.nw) di += R.DIl[d * R.sDIl + (size_t)(t.di_base + w) * 16 + (lx & (TT_LB - 1))];
            }
            if (x >= t.j0 && x < t.j0 + t.nj)
                for (int w = 0; w < t.nw; ++w) jd += R.Jd[d * R.sJd + t.jd_base + (size_t)w * t.nj + (x - t.j0)];
        }
        R.Di[d * R.sO + (size_t)iI * N + x] = di;
        R.JD[d * R.sO + (size_t)iI * N + x] = jd;
    }
}

__global__ __launch_bounds__(TT_RED_THREADS) void jk_tile_reduce_kernel(TRArgs R)
{
    __shared__ double sPart[TT_RED_THREADS];
    const int N = R.N;
    const int n0 = N * R.xtiles, n1 = R.jt_rows * R.xtiles, n2 = N;
    int b = blockIdx.x;
    const int d = b / (n0 + n1 + n2);
    b -= d * (n0 + n1 + n2);
    if (b < n0) {
        // 64-column tiles of the internal index space that do not straddle a class: tile t of class X starts at cstart[X] + 64 t'
        const int jI = b / R.xtiles, tl = b % R.xtiles;
        int X0 = -1, cnt = 0;
        for (int c = 0; c < 4 && X0 < 0; ++c) {
            const int nt = (R.tab[TVT_CSIZE + c] + 63) / 64;
            if (tl < cnt + nt) X0 = R.tab[TVT_CSTART + c] + 64 * (tl - cnt);
            cnt += nt;
        }
        if (X0 >= 0) tr_dj_block(R, d, jI, X0, sPart);
    } else if (b < n0 + n1) {
        b -= n0;
        tr_jt_block(R, d, b / R.xtiles, b % R.xtiles, sPart);
    } else
        tr_gather_block(R, d, b - n0 - n1);
}

// Original indices (x, y): K = D + D2^T with D = Dj + Di + ED + EDT^T (D2 = D for a symmetric density; a general one: D = D(P^T), D2 = D(P));
// J[x][y] = JD + EJ of the pair (hi, lo) + the Jt total of the pair
struct TFArgs {
    const double *Dj, *Di, *ED, *EDT, *Dj2, *Di2, *ED2, *EDT2, *JD, *EJ, *JtTot;
    const int *tab;               // TView::tab
};
__global__ void jk_tile_final_kernel(TFArgs F, BLayout L, double *__restrict__ J, double *__restrict__ K)
{
    const int N = L.N;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int x = e / N, y = e - x * N;
    const int wx = L.ao[x], wy = L.ao[y];
    const size_t sx = ao_sigma(L, wx), sy = ao_sigma(L, wy);
    const size_t xy = sx * N + sy, yx = sy * N + sx;
    K[e] = (F.Dj[xy] + F.Di[xy] + F.ED[xy] + F.EDT[yx]) + (F.Dj2[yx] + F.Di2[yx] + F.ED2[yx] + F.EDT2[xy]);
    const int whi = x >= y ? wx : wy, wlo = x >= y ? wy : wx;                // the pair (hi >= lo) in original order
    const size_t hl = (x >= y) ? xy : yx;
    const int p = F.tab[TVT_PID + ao_cls(whi) * 4 + ao_cls(wlo)];
    const bool rows_hi = ao_cls(whi) == F.tab[TVT_PA + p];                             // the pair matrix is [class a][class b]; a triangle holds (k >= l): hi first
    const int kr = rows_hi ? ao_loc(whi) : ao_loc(wlo), lc = rows_hi ? ao_loc(wlo) : ao_loc(whi);
    J[e] = F.JD[hl] + F.EJ[hl] + F.JtTot[F.tab[TVT_PMOFF + p] + (size_t)kr * F.tab[TVT_PMPITCH + p] + lc];
}
