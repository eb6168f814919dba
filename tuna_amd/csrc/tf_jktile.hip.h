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

// The stored part of local rows as symmetric matrices, for the GEMM-shaped consumers (AO->MO): the tiles counterpart of
// unpack_own_rows_blocked_kernel (tf_jkpacked.hip.h).
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
    double *DJ, *Jt, *Jd, *DIk, *DIl;   // DJ and Jd: density-minor ([index][ND]: written at every step); the others: one plane per density
    size_t sJt, sDIk, sDIl;             // strides between the densities of a pass
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
#ifndef TT_STEPS_MAX
#define TT_STEPS_MAX 48
#endif
#define TT_STEPS_MAX_DOC                          // steps of a task at most (tf_tiles_host.h: part_steps): its density rows are staged in LDS once
#define TT_TRP 17
template <int ND, int MB>
struct TJLds {
    static constexpr bool STAGE = ND <= 2;       // one or two densities: their rows of every step and the Jd weights live in LDS; four and
                                                 // more: the rows come through the prefetch ring (lanes = densities), the weights sit in registers
    double slots[TT_KB * TT_W * ND * 16 * MB];   // row sums of a block's steps: [step][wave][density][row]
    double2 pp[STAGE ? TT_W : 1][STAGE ? ND : 1][MB * 2][STAGE ? 64 : 1];
    double jdl[STAGE ? TT_W : 1][STAGE ? ND : 1][TT_KB][STAGE ? 64 : 1];   // Jd: the lanes' partial sums of a block's steps (summed over the wave four steps at a time)
    double tr[ND > 1 ? TT_W : 1][ND > 1 ? MB : 1][ND > 1 ? 16 * TT_TRP : 1];   // (ND > 1) a wave's 16 x 16 blocks of the current slice, [k][l] (pitch TT_TRP: odd, no bank conflicts; one buffer per
                                                 // block: a single one serialises the blocks on the LDS round trip -- 2.1 -> 2.9 ms): written
                                                 // from the A-operand registers, read back transposed (lane = column l) for the sums over k
    // the density rows of ALL steps of the task, staged by the whole workgroup in the prologue (nothing is loaded for them inside the loop:
    // a wait for such a load would drain the prefetched slices): P_d[j][k] of the task's rows, P_d[j][l] of its column blocks, Pp_d[ij]
    double pk[STAGE ? ND : 1][STAGE ? TT_STEPS_MAX : 1][16 * MB];
    double pl[STAGE ? ND : 1][STAGE ? TT_STEPS_MAX : 1][TT_W * TT_LB];
    double pij[ND][TT_STEPS_MAX];
};

#define TJ_U(x) __builtin_amdgcn_readfirstlane(x)      // wave-uniform: keep it in a scalar register for the whole task
template <int ND, int MB, int PF, bool DIAG>
__device__ __forceinline__ void tj_run(const double *__restrict__ T, const double *__restrict__ X, const double *__restrict__ Pm, const TJArgs &A,
                                       const TTask *__restrict__ tp, int w, int lane, bool active, TJLds<ND, MB> &S)
{
    const int N = A.N;
    const int m = lane & 15, kk = lane >> 4;
    // the task record, once, into scalar registers (a field read inside the loop is a scalar load and a wait per step)
    const int ta = TJ_U(tp->a), tb = TJ_U(tp->b), k0 = TJ_U(tp->k0), nks = TJ_U(tp->nks), roff0 = TJ_U(tp->roff0), lb0 = TJ_U(tp->lb0), tnw = TJ_U(tp->nw);
    const int tnl = TJ_U(tp->nl), slice = TJ_U(tp->slice), woffw = TJ_U(tp->woff[w]), jt_pitch = TJ_U(tp->jt_pitch), dj_len = TJ_U(tp->dj_len);
    const int dj_koff = TJ_U(tp->dj_koff), dj_loffw = TJ_U(tp->dj_loff[w]), di_base = TJ_U(tp->di_base), jd_base = TJ_U(tp->jd_base);
    const int self_last = TJ_U(tp->self_last), kbase = TJ_U(tp->kbase), lbase = TJ_U(tp->lbase), ncol = TJ_U(tp->ncol), pm_off = TJ_U(tp->pm_off);
    const int pm_pitch = TJ_U(tp->pm_pitch), iI = TJ_U(tp->i), j0 = TJ_U(tp->j0), nj = TJ_U(tp->nj);
    const long long tbase = tp->base, jt_base = tp->jt_base, dj_base = tp->dj_base;
    const bool tri = ta == tb;
    const int lb = lb0 + w, ks = k0 / TT_KS;
    const int nwg = (int)(blockDim.x >> 6);
    const size_t nn = (size_t)N * N;
    // ---- per-lane constants
    unsigned offA[MB], offB[MB];
    int rdiag[MB];
    bool rv[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int r = 16 * mb + m;
        rv[mb] = active && r < nks;
        const int rs = roff0 + r;
        const int rl = rv[mb] ? tt_row_len(tri, ks, lb, rs, tnl) : 0;
        const int nst = tt_min(TT_KS, TJ_U(tp->nk) - TT_KS * ks);          // rows of the STORED strip (the chunk order inside a block of 16 rows depends on it)
        offA[mb] = (4 * kk < rl) ? 8u * (unsigned)(woffw + tt_elem_off(tri, ks, lb, rs, 4 * kk, nst, tnl)) : TF_BUF_OOB;
        offB[mb] = (4 * kk + 2 < rl) ? 8u * (unsigned)(woffw + tt_elem_off(tri, ks, lb, rs, 4 * kk + 2, nst, tnl)) : TF_BUF_OOB;
        rdiag[mb] = (k0 + r) - (TT_LB * lb + 4 * kk);                    // the register of this lane that holds k == l (triangles)
    }
    const int lcol0 = TT_LB * lb + 4 * kk;                                 // loc of the lane's first column
    // buffer descriptors: the task's region of the tensor; its DJ vectors (row of step s at s * dj_len); this wave's Jd values
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(T + tbase), 0, (int)min((long long)nj * slice * 8, 0x7fffffffLL), 0x00020000);
    constexpr bool R4V = ND == 1;                                           // the column sums on the vector unit (one density) or the matrix core
    const __amdgpu_buffer_rsrc_t rdj = __builtin_amdgcn_make_buffer_rsrc(A.DJ + dj_base * ND, 0, (int)min((long long)nj * dj_len * ND * 8, 0x7fffffffLL), 0x00020000);
    const __amdgpu_buffer_rsrc_t rjd = __builtin_amdgcn_make_buffer_rsrc(A.Jd + ((size_t)jd_base + (size_t)w * nj) * ND, 0, nj * ND * 8, 0x00020000);
    const unsigned st_l = R4V ? (((m & 3) == 0) ? 8u * (unsigned)(dj_loffw + 4 * kk + (m >> 2)) : TF_BUF_OOB)   // one density: lanes m = 0, 4, 8, 12 of every row
                              : ((m < ND) ? 8u * (unsigned)((dj_loffw + kk) * ND + m) : TF_BUF_OOB);            // lane (density m, row kk), registers q: rows kk + 4 q
    const unsigned st_jd = ((lane & 15) == 0) ? 8u * (unsigned)((lane >> 4) * ND) : TF_BUF_OOB;
    // Column sums D_d[j][l] = sum_k m P_d[i][k].  One density: lane-local products and a transposing butterfly over the 16 lanes of a DPP row
    // (measured at N = 400: 2.2 ms against 2.9 through the matrix core, whose LDS round trip per block the single density does not pay
    // for); several: on the matrix core from the transposed blocks (lane = (column d, row group g), K index 4 q + g), B operand below.
    double pik[R4V ? MB : 1];
#pragma unroll
    for (int mb = 0; mb < (R4V ? MB : 1); ++mb) pik[mb] = (R4V && rv[mb]) ? X[(size_t)iI * N + kbase + k0 + 16 * mb + m] : 0.0;
    double b4[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * mb + 4 * q + kk;
            b4[mb][q] = (active && m < ND && r < nks) ? X[(size_t)(m < ND ? m : 0) * nn + (size_t)iI * N + kbase + k0 + r] : 0.0;
        }
    // B operand of the row sums: column d = P_d[i][l] (lanes m == d), zero elsewhere
    double b1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[r] = (active && m < ND && lcol0 + r < ncol) ? X[(size_t)(m < ND ? m : 0) * nn + (size_t)iI * N + lbase + lcol0 + r] : 0.0;
    // weights of Jd: Pp_d(k, l) of the lane's elements (pad columns of the pair matrices are zero): in LDS, or in registers (ND >= 4)
    constexpr bool STAGE = TJLds<ND, MB>::STAGE;
    double ppr[STAGE ? 1 : ND][MB][4];
    if (active) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const double *src = Pm + (size_t)d * A.pm_len + pm_off + (size_t)(k0 + 16 * mb + m) * pm_pitch + lcol0;
                const double2 q0 = (rv[mb] && lcol0 < pm_pitch) ? *reinterpret_cast<const double2 *>(src) : make_double2(0.0, 0.0);
                const double2 q1 = (rv[mb] && lcol0 + 2 < pm_pitch) ? *reinterpret_cast<const double2 *>(src + 2) : make_double2(0.0, 0.0);
                if constexpr (STAGE) { S.pp[w][d][2 * mb][lane] = q0; S.pp[w][d][2 * mb + 1][lane] = q1; }
                else { ppr[d][mb][0] = q0.x; ppr[d][mb][1] = q0.y; ppr[d][mb][2] = q1.x; ppr[d][mb][3] = q1.y; }
            }
    } else if constexpr (!STAGE) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) ppr[d][mb][r] = 0.0;
    }
    // accumulators.  D_d[i][k] and D_d[i][l] -- sums over the steps -- are lane-local multiply-adds for one or two densities (M25 false)
    // and two more products on the matrix core (columns = densities, accumulated over the steps) for four and more.
    constexpr bool M25 = ND >= 4;
    double jt[ND][MB][4], x5[M25 ? 1 : ND][4], x2[M25 ? 1 : ND][MB];
    tt_v4d acc2[MB], acc5 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc2[mb] = tt_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int d = 0; d < (M25 ? 1 : ND); ++d) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x5[d][r] = 0.0;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) x2[d][mb] = 0.0;
    }
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) jt[d][mb][r] = 0.0;
    // the density rows of the task's steps -> LDS (all threads of the workgroup; the barrier in front of the loop follows), a group of
    // loads in flight per thread (a loop of load - wait - store was 10 us of a 40 us task).
    {
        const int nthr = (int)blockDim.x, tid = (int)threadIdx.x;
        constexpr int G = 8;                                                // loads in flight per thread
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double *Xd = X + d * nn;
            if constexpr (STAGE) {
                for (int e0 = tid; e0 < nj * 16 * MB; e0 += G * nthr) {
                    double v[G];
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int e = e0 + u * nthr, sq = e / (16 * MB), x = e - sq * (16 * MB);
                        v[u] = (sq < nj && x < nks) ? Xd[(size_t)(j0 + sq) * N + kbase + k0 + x] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int e = e0 + u * nthr;
                        if (e < nj * 16 * MB) (&S.pk[d][0][0])[e] = v[u];
                    }
                }
                for (int e0 = tid; e0 < nj * TT_W * TT_LB; e0 += G * nthr) {
                    double v[G];
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int e = e0 + u * nthr, sq = e / (TT_W * TT_LB), x = e - sq * (TT_W * TT_LB), lc = TT_LB * lb0 + x;
                        v[u] = (sq < nj && lc < ncol) ? Xd[(size_t)(j0 + sq) * N + lbase + lc] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < G; ++u) {
                        const int e = e0 + u * nthr;
                        if (e < nj * TT_W * TT_LB) (&S.pl[d][0][0])[e] = v[u];
                    }
                }
            }
            if (tid < nj) {
                const int jI = j0 + tid;
                const double xij = Xd[(size_t)iI * N + jI];
                S.pij[d][tid] = (jI == iI) ? xij : xij + Xd[(size_t)jI * N + iI];
            }
        }
    }
    // ND >= 4: the B operands of the sums over the steps -- P_d[j][l] of the wave's columns, P_d[j][k] of its rows, d = lane & 15 -- come
    // through the ring with the slices (loads of the lanes m < ND; a step beyond the last: out of range = zeros)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(X), 0, (int)min((long long)ND * N * N * 8, 0x7fffffffLL), 0x00020000);
    const unsigned ax_l0 = (active && m < ND && lcol0 < ncol) ? 8u * (unsigned)(m * N * N + lbase + lcol0) : TF_BUF_OOB;
    const unsigned ax_l1 = (active && m < ND && lcol0 + 2 < ncol) ? 8u * (unsigned)(m * N * N + lbase + lcol0 + 2) : TF_BUF_OOB;
    unsigned ax_k[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 16 * mb + 4 * q + kk;
            ax_k[mb][q] = (active && m < ND && r < nks) ? 8u * (unsigned)(m * N * N + kbase + k0 + r) : TF_BUF_OOB;
        }
    struct TJAux { double2 l[2]; double k[MB][4]; };
    auto load_aux = [&](TJAux &Q, int s) {                                   // (s >= nj: nothing)
        const unsigned so = 8u * (unsigned)((j0 + min(s, nj - 1)) * N);
        const bool ok = s < nj;
        Q.l[0] = buf_load2<0>(rx, ok ? ax_l0 : TF_BUF_OOB, so);
        Q.l[1] = buf_load2<0>(rx, ok ? ax_l1 : TF_BUF_OOB, so);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int q = 0; q < 4; ++q) Q.k[mb][q] = buf_load1<0>(rx, ok ? ax_k[mb][q] : TF_BUF_OOB, so);
    };
    // The wave's pieces of the next PF slices live in a ring of PF register sets: the loads of row block mb of step s + PF are issued as
    // soon as row block mb of step s has been consumed (the last steps re-load the last slice: no branch, no copies between the sets).
    auto load_mb = [&](double2 (&B)[MB][2], int mb, int s) {
        const unsigned so = (unsigned)s * (unsigned)slice * 8u;
        B[mb][0] = buf_load2<2>(rt, offA[mb], so);
        B[mb][1] = buf_load2<2>(rt, offB[mb], so);
    };
    auto step = [&](double2 (&B)[MB][2], TJAux &Q, int s, int kq) {
        const bool self = self_last && s == nj - 1;
        const int sc = min(s, nj - 1);                                      // (every step of a block runs -- those beyond the last on the last step's
        const double live = s < nj ? 1.0 : 0.0;                             // data with zero weights, their stores out of range: a branch around a
        const int snext = min(s + PF, nj - 1);                              // step would make the ring a set of copies)
        double ppij[ND], pjl[M25 ? 1 : ND][4], pjk[M25 ? 1 : ND][MB], b2[4];
#pragma unroll
        for (int d = 0; d < ND; ++d) ppij[d] = live * S.pij[d][sc];
        if constexpr (M25) {
            // B operand of D_d[i][k] += sum_l m P_d[j][l]: column d = lanes m == d (from the ring)
            b2[0] = Q.l[0].x; b2[1] = Q.l[0].y; b2[2] = Q.l[1].x; b2[3] = Q.l[1].y;
        } else {
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const double2 q0 = *reinterpret_cast<const double2 *>(&S.pl[d][sc][TT_LB * w + 4 * kk]), q1 = *reinterpret_cast<const double2 *>(&S.pl[d][sc][TT_LB * w + 4 * kk + 2]);
                pjl[d][0] = live * q0.x; pjl[d][1] = live * q0.y; pjl[d][2] = live * q1.x; pjl[d][3] = live * q1.y;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) pjk[d][mb] = live * S.pk[d][sc][16 * mb + m];
            }
        }
        double jd[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) jd[d] = 0.0;
        tt_v4d accs[MB];
        tt_v4d acc4 = {0.0, 0.0, 0.0, 0.0};                                 // column sums of the step: rows l, columns d
        double t4[4] = {0.0, 0.0, 0.0, 0.0};
        double *trw = &S.tr[R4V ? 0 : w][0][0];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const double tv[4] = {B[mb][0].x, B[mb][0].y, B[mb][1].x, B[mb][1].y};
            // the block into LDS, [k = m][l]: the lane's two column pairs
            if constexpr (!R4V) {
                double *dst = trw + mb * (16 * TT_TRP) + m * TT_TRP + 4 * kk;
                dst[0] = tv[0]; dst[1] = tv[1]; dst[2] = tv[2]; dst[3] = tv[3];
            }
            tt_v4d acc = {0.0, 0.0, 0.0, 0.0};
#ifndef TJ_ABL_NOMFMA                    // (TJ_ABL_*: timing experiments only, wrong results -- tools/build_variant.sh)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[r], b1[r], acc, 0, 0, 0);
#endif
            accs[mb] = acc;
            if constexpr (M25) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc2[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(tv[r], b2[r], acc2[mb], 0, 0, 0);
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                double ppv[4];
                if constexpr (STAGE) {
                    const double2 w0 = S.pp[w][d][2 * mb][lane], w1 = S.pp[w][d][2 * mb + 1][lane];
                    ppv[0] = w0.x; ppv[1] = w0.y; ppv[2] = w1.x; ppv[3] = w1.y;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ppv[r] = ppr[d][mb][r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double tm = tv[r];
#ifdef TJ_ABL_NOVALU
                    jd[d] += tm;
#else
                    jt[d][mb][r] += tm * ppij[d];
                    jd[d] += tm * ppv[r];
                    if constexpr (!M25) {
                        const double to = (DIAG && r == rdiag[mb]) ? 0.0 : tm;      // without the diagonal k == l
                        x2[d][mb] += tm * pjl[d][r];
                        x5[d][r] += to * pjk[d][mb];
                        if constexpr (R4V) t4[r] += to * pik[mb];
                    }
#endif
                }
            }
            // the block back, transposed: lane (column l = m, row group kk) holds the rows k = 4 q + kk -- the A operand of the sums over k
#ifndef TJ_ABL_NOR4
            if constexpr (!R4V) {
                const double *src = trw + mb * (16 * TT_TRP) + kk * TT_TRP + m;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    double tw = src[4 * q * TT_TRP];
                    if (DIAG && (k0 + 16 * mb + 4 * q + kk) == (TT_LB * lb + m)) tw = 0.0;      // without the diagonal k == l
                    acc4 = __builtin_amdgcn_mfma_f64_16x16x4f64(tw, b4[mb][q], acc4, 0, 0, 0);
                    if constexpr (M25) {                                    // D_d[i][l] += sum_k m P_d[j][k]: B column d = lanes m == d, K index 4 q + kk
                        acc5 = __builtin_amdgcn_mfma_f64_16x16x4f64(tw, Q.k[mb][q], acc5, 0, 0, 0);
                    }
                }
            }
#endif
            // the slice of step s + PF into the registers this row block has just been read from (issued behind their last use: the
            // ring positions stay the same physical registers around the loop -- no copies, no wait for the loads in flight)
            __builtin_amdgcn_sched_barrier(0);
            load_mb(B, mb, snext);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (M25) {
            __builtin_amdgcn_sched_barrier(0);
            load_aux(Q, s + PF);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (STAGE) {
#pragma unroll
            for (int d = 0; d < ND; ++d) S.jdl[w][d][kq][lane] = jd[d];
        } else {
            // Jd of the step: the wave totals of the ND densities together; density-minor store
            double v;
            if constexpr (ND == 4) v = wave_sum4(jd[0], jd[1], jd[2], jd[3]);          // lane L: density L >> 4
            else { const double v8[8] = {jd[0], jd[1], jd[2], jd[3], jd[4 % ND], jd[5 % ND], jd[6 % ND], jd[7 % ND]}; v = wave_sum8(v8); }   // lane L: density L >> 3
            constexpr int SH = ND == 4 ? 4 : 3;
            const bool own = (lane & ((1 << SH) - 1)) == 0 && s < nj;
            buf_store1<0>(rjd, own ? 8u * (unsigned)(lane >> SH) : TF_BUF_OOB, (unsigned)(min(s, nj - 1) * ND * 8), v);
        }
        // column sums D_d[j][l]: column d of acc4 = lanes m == d, rows l = kk + 4 q
        {
            // (the row (i, i) takes no D[j][.] terms; a step beyond the last stores nothing.  Dropped through the LANE offset: the scalar
            // offset of a buffer access is not range-checked)
            const unsigned lo = (self || s >= nj) ? TF_BUF_OOB : st_l;
            const unsigned srow = (unsigned)min(s, nj - 1) * (unsigned)dj_len * (unsigned)(8 * ND);
#ifndef TJ_ABL_NOR4
            if constexpr (R4V) buf_store1<0>(rdj, lo, srow, row_sum4(t4[0], t4[1], t4[2], t4[3]));   // lane m holds column 4 kk + (m >> 2)
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) buf_store1<0>(rdj, lo, srow + (unsigned)(32 * ND * q), acc4[q]);
            }
#endif
        }
        // row sums D_d[j][k]: column d of the result tiles = lanes m == d, rows kk + 4 q (written behind the vector work: the matrix
        // core has long delivered)
        if (m < ND) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                double *sl = S.slots + ((size_t)(kq * TT_W + w) * ND + m) * (16 * MB) + mb * 16 + kk * 4;
                *reinterpret_cast<double2 *>(sl) = make_double2(accs[mb][0], accs[mb][1]);
                *reinterpret_cast<double2 *>(sl + 2) = make_double2(accs[mb][2], accs[mb][3]);
            }
        }
    };
    double2 R[PF][MB][2];
    TJAux RQ[PF];
    if (active) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) load_mb(R[q], mb, min(q, nj - 1));
            if constexpr (M25) load_aux(RQ[q], q);
        }
    }
    __syncthreads();                                                       // the density rows and the Jd weights are in place
    for (int sb = 0; sb < nj; sb += TT_KB) {
        if (active) {
            // (the steps of a block are a LOOP: unrolled, the compiler overlaps them and needs twice the registers; PF steps per
            // iteration keep the ring positions compile-time constants)
#pragma unroll 1
            for (int kq0 = 0; kq0 < TT_KB; kq0 += PF) {
#pragma unroll
                for (int q = 0; q < PF; ++q) step(R[q], RQ[q], sb + kq0 + q, kq0 + q);
            }
            // Jd of the steps sb .. sb + 3: the wave totals of four values together (steps beyond the last: out of the buffer's range)
            if constexpr (STAGE)
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const double v = wave_sum4(S.jdl[w][d][0][lane], S.jdl[w][d][1][lane], S.jdl[w][d][2][lane], S.jdl[w][d][3][lane]);
                buf_store1<0>(rjd, ((lane & 15) == 0 && sb + (lane >> 4) < nj) ? st_jd : TF_BUF_OOB, (unsigned)((sb * ND + d) * 8), v);
            }
        }
#ifndef TJ_ABL_NOMERGE
        jkp_lds_barrier();
#endif
        // merge of the row sums: wave w adds the waves' partials of step sb + w (+ nwg ..) and writes the K slot of that row
        for (int kq = w; kq < TT_KB; kq += nwg) {
            const int s = sb + kq;
            if (s >= nj) break;
            const bool self = self_last && s == nj - 1;
            // the 16 MB rows x ND densities of the step, density-minor as they are stored: element e = row ND + d
            for (int e = lane; e < 16 * MB * ND; e += 64) {
                const int row = e / ND, d = e - row * ND;                  // row: position in the slots = mb 16 + kk 4 + q  <->  strip row mb 16 + kk + 4 q
                double v = S.slots[((size_t)(kq * TT_W) * ND + d) * (16 * MB) + row];
                for (int u = 1; u < tnw; ++u) v += S.slots[((size_t)(kq * TT_W + u) * ND + d) * (16 * MB) + row];
                const int krel = 16 * (row >> 4) + ((row >> 2) & 3) + 4 * (row & 3);
                buf_store1<0>(rdj, (krel < nks && !self) ? 8u * (unsigned)((dj_koff + krel) * ND + d) : TF_BUF_OOB, (unsigned)s * (unsigned)dj_len * (unsigned)(8 * ND), v);
            }
        }
#ifndef TJ_ABL_NOMERGE
        jkp_lds_barrier();
#endif
    }
    if (!active) return;
    // ---- what was summed over the steps
#pragma unroll
    for (int d = 0; d < ND; ++d) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {                                   // the Jt tile
            double *dst = A.Jt + d * A.sJt + jt_base + (long long)(k0 + 16 * mb + m) * jt_pitch + lcol0;
            if (rv[mb] && lcol0 < jt_pitch) *reinterpret_cast<double2 *>(dst) = make_double2(jt[d][mb][0], jt[d][mb][1]);
            if (rv[mb] && lcol0 + 2 < jt_pitch) *reinterpret_cast<double2 *>(dst + 2) = make_double2(jt[d][mb][2], jt[d][mb][3]);
        }
        if constexpr (!M25) {
            const double cs = row_sum4(x5[d][0], x5[d][1], x5[d][2], x5[d][3]);   // D[i][l]: over the rows
            if ((m & 3) == 0) A.DIl[d * A.sDIl + (size_t)(di_base + w) * 16 + 4 * kk + (m >> 2)] = cs;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {                               // D[i][k]: over the four lane groups
                double v = x2[d][mb];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if (kk == 0) A.DIk[d * A.sDIk + (size_t)(di_base + w) * 64 + 16 * mb + m] = v;
            }
        }
    }
    if constexpr (M25) {                                                    // column d of the result tiles = lanes m == d, rows kk + 4 q
        if (m < ND) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                A.DIl[m * A.sDIl + (size_t)(di_base + w) * 16 + kk + 4 * q] = acc5[q];
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) A.DIk[m * A.sDIk + (size_t)(di_base + w) * 64 + 16 * mb + kk + 4 * q] = acc2[mb][q];
            }
        }
    }
}

// One workgroup per task; blockDim = 64 x (waves of the launch's bucket: >= the task's column blocks).  MB = row blocks of 16 of a
// task's strip (4: strips of 64 rows, one density; fewer for the passes over several densities: tf_tiles.h, `ksub`).
// (eight densities: 256 VGPRs + 61 AGPRs, one wave per SIMD.  Compiled for two -- 28 spilled registers with a one-slot ring, 331 with
// two -- the kernel takes 9.2 / 18.8 ms against 8.6: occupancy is not what it lacks)
template <int ND, int MB, int PF>
__global__ __launch_bounds__(64 * TT_W, (MB >= 4) ? 1 : (ND >= 8 ? 1 : 2)) void jk_tile_kernel(const double *__restrict__ T, const TTask *__restrict__ tasks,
                                                                            const double *__restrict__ X, const double *__restrict__ Pm, TJArgs A)
{
    static_assert(TT_KB == 4 && TT_KB % PF == 0, "the ring position of a step must be a compile-time constant");
    __shared__ TJLds<ND, MB> S;
    const TTask *__restrict__ tp = tasks + blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool active = w < tp->nw;
    // a triangle's tile holds diagonal elements k == l only where its rows and columns overlap
    const int lb = tp->lb0 + w;
    const bool diag = tp->a == tp->b && active && TT_LB * lb <= tp->k0 + tp->nks - 1 && TT_LB * lb + TT_LB - 1 >= tp->k0;
    if (diag) tj_run<ND, MB, PF, true>(T, X, Pm, A, tp, w, lane, active, S);
    else tj_run<ND, MB, PF, false>(T, X, Pm, A, tp, w, lane, active, S);
}

// ---- the edge elements m = (ij|il), l <= j (k == i): one workgroup per first index i ---------------------------------------------------
// Per-i outputs, internal indices: EJ[i][x] (J of the pair (i, x)), ED[i][x] (D[i][x]), EDT[i][x] (D[x][i]).  The term D[j][l] += w m P[i][i]
// runs over i and is added by the D[j][.] waves of jk_tile_reduce_kernel.  A wave takes every fourth row j of a class, its lanes the
// columns l <= j (coalesced reads of E[j][.]); sums over l: over the wave; sums over j: in the wave's own copy of the output row in LDS;
// the four copies are added in fixed order: bitwise reproducible.
struct TEArgs {
    long long edge_base;
    int N;
    const int *tab;               // TView::tab
    double *EJ, *ED, *EDT;
    size_t sE;                    // stride between densities
};
#define TT_EDGE_WAVES 4
template <int ND>
__global__ __launch_bounds__(64 * TT_EDGE_WAVES) void jk_edge_kernel(const double *__restrict__ T, const double *__restrict__ X, const TRunI *__restrict__ runs, TEArgs A)
{
    extern __shared__ double sE[];                                         // [Xi N | Xti N | per wave: J N, D N, DT N]
    const int N = A.N, iI = blockIdx.x;
    const size_t nn = (size_t)N * N;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *Xi = sE, *Xti = sE + N, *accJ = sE + 2 * N + (size_t)w * 3 * N, *accD = accJ + N, *accT = accD + N;
    {
        const int d = blockIdx.y;                                           // grid (N, densities of the pass)
        const double *Xd = X + d * nn;
        for (int x = threadIdx.x; x < N; x += blockDim.x) { Xi[x] = Xd[(size_t)iI * N + x]; Xti[x] = Xd[(size_t)x * N + iI]; }
        for (int x = lane; x < 3 * N; x += 64) accJ[x] = 0.0;
        __syncthreads();
        const double xii = Xi[iI];
        double dii = 0.0;                                                  // this lane's share of D[i][i]
        for (int cj = 0; cj < 4; ++cj) {
            const TRunI R = runs[(size_t)iI * 4 + cj];
            if (R.nj <= 0) continue;
            const int c0 = A.tab[TVT_CSTART + cj], l0 = R.j0 - c0;
            const double *E = T + A.edge_base + R.e_base - (long long)l0 * (l0 + 1) / 2;     // E[T(loc j) + loc l]
            for (int lj = l0 + w; lj < l0 + R.nj; lj += TT_EDGE_WAVES) {   // row (i, j)
                const int jI = c0 + lj;
                const double *Ej = E + (long long)lj * (lj + 1) / 2;
                const double ppij = (jI == iI) ? xii : Xi[jI] + Xti[jI], xji = Xti[jI];
                double ej = 0.0, edt = 0.0;
                for (int ll = lane; ll <= lj; ll += 64) {
                    const int lI = c0 + ll;
                    const double mv = Ej[ll], wgt = (ll == lj) ? 0.5 : 1.0;
                    const double ppil = (lI == iI) ? xii : Xi[lI] + Xti[lI];
                    ej += mv * ppil;                                       // Jd[ij] += m Pp[il]
                    dii += wgt * mv * Xd[(size_t)jI * N + lI];             // D[i][i] += w m P[j][l]
                    edt += wgt * mv * Xi[lI];                              // D[j][i] += w m P[i][l]   (i != j)
                    if (ll != lj) accJ[lI] += mv * ppij;                   // Jt[il] += m Pp[ij]       (l != j)
                    if (lI != iI) accD[lI] += wgt * mv * xji;              // D[i][l] += w m P[j][i]   (l != i)
                }
                const double s4 = wave_sum4(ej, edt, 0.0, 0.0);            // lane L: total of value L >> 4
                if (lane == 0) accJ[jI] += s4;                             // (behind this wave's own column updates: one wave, program order)
                if (lane == 16 && jI != iI) accT[jI] += s4;
            }
        }
        dii = wave_sum1(dii);
        if (lane == 0) accD[iI] += dii;
        __syncthreads();
        for (int x = threadIdx.x; x < N; x += blockDim.x) {
            double j = 0.0, dd = 0.0, dt = 0.0;
            for (int u = 0; u < TT_EDGE_WAVES; ++u) { const double *q = sE + 2 * N + (size_t)u * 3 * N; j += q[x]; dd += q[N + x]; dt += q[2 * N + x]; }
            A.EJ[d * A.sE + (size_t)iI * N + x] = j;
            A.ED[d * A.sE + (size_t)iI * N + x] = dd;
            A.EDT[d * A.sE + (size_t)iI * N + x] = dt;
        }
        __syncthreads();
    }
}

// ---- reductions: one launch, three kinds of workgroups (4 waves), fixed summation order everywhere ------------------------------
struct TRArgs {
    const int *itask_ptr, *itasks, *jlist_ptr;
    const int *clsI;
    const double *DJ, *Jt, *Jd, *DIk, *DIl, *T, *X;
    size_t sJt, sDIk, sDIl, sO;                 // (DJ and Jd are density-minor: [index][nd])
    double *Dj, *Di, *JtTot, *JD; // Dj[j][x], Di[i][x], JD[i][j]: [N][N] internal; JtTot: pair matrices
    long long edge_base;
    int N, ksub, npair, nd, pm_len;
    const int *tab;               // TView::tab
    int jt_rows;                  // rows of all pair matrices together
};
#define TT_RED_THREADS 512
#define TT_RED_WAVES 8
#define TT_RED_CT 4               // column tiles of 64 a wave keeps in registers (classes of up to 256 AOs per pass)

// the waves' partial rows -> the row: wave 0 adds them in fixed order (sPart: [TT_RED_WAVES][64 TT_RED_CT])
__device__ __forceinline__ void tr_combine(double *sPart, const double (&acc)[TT_RED_CT], int w, int lane, double (&tot)[TT_RED_CT])
{
#pragma unroll
    for (int u = 0; u < TT_RED_CT; ++u) sPart[((size_t)w * TT_RED_CT + u) * 64 + lane] = acc[u];
    __syncthreads();
    if (w == 0) {
#pragma unroll
        for (int u = 0; u < TT_RED_CT; ++u) {
            double t = 0.0;
            for (int q = 0; q < TT_RED_WAVES; ++q) t += sPart[((size_t)q * TT_RED_CT + u) * 64 + lane];
            tot[u] = t;
        }
    }
    __syncthreads();
}

// kind 0, one workgroup per (j, class X): D[j][x] for the AOs x of class X: the DJ vectors of the rows (i, j), i != j (jlist) + the edge
// term; the waves share the rows.
__device__ __forceinline__ void tr_dj_block(const TRArgs &R, const TPairI *__restrict__ pairs, const TRunI *__restrict__ runs, const int *__restrict__ jlist,
                                            int d, int jI, int cX, int w, int lane, double *sPart)
{
    const int N = R.N;
    const int cj = R.clsI[jI];
    const int xs = R.tab[TVT_CSTART + cX], nX = R.tab[TVT_CSIZE + cX];
    const int lj = jI - R.tab[TVT_CSTART + cj];
    const double *DJ = R.DJ + d;                                           // density-minor
    const int nd = R.nd;
    const double *Xd = R.X + (size_t)d * N * N;
    for (int t0 = 0; t0 < nX; t0 += 64 * TT_RED_CT) {
        double acc[TT_RED_CT], tot[TT_RED_CT];
#pragma unroll
        for (int u = 0; u < TT_RED_CT; ++u) acc[u] = 0.0;
        for (int q = R.jlist_ptr[jI] + w; q < R.jlist_ptr[jI + 1]; q += TT_RED_WAVES) {
            const int iI = jlist[q];
            const int c = R.clsI[iI] ^ cj;
            const int p = R.tab[TVT_PID + cX * 4 + (cX ^ c)];
            const TPairI P = pairs[(size_t)iI * 10 + p];
            const TRunI Rn = runs[(size_t)iI * 4 + cj];
            const double *vec = DJ + (Rn.dj_base + (long long)(jI - Rn.j0) * Rn.dj_len) * nd;
            const int a = R.tab[TVT_PA + p], b = R.tab[TVT_PB + p];
            const bool tri = a == b, have = P.first_task >= 0;
            const long long l0 = Rn.j0 - R.tab[TVT_CSTART + cj];
            const double *Ee = R.T + R.edge_base + Rn.e_base + (long long)lj * (lj + 1) / 2 - l0 * (l0 + 1) / 2;
            const double xii = Xd[(size_t)iI * N + iI];
#pragma unroll
            for (int u = 0; u < TT_RED_CT; ++u) {
                const int lx = t0 + 64 * u + lane;
                double v = 0.0;
                if (have && cX == a && lx < P.nk) {                       // x as a row index k: the chunks of its stored strip
                    const int ks = lx / TT_KS, nst = min(TT_KS, P.nk - TT_KS * ks);
                    int nch, wv;
                    tt_chunks(tt_nlb(tri, ks, P.nk, P.nl), &nch, &wv);
                    const double *src = vec + (size_t)(P.dj_k + tt_dj_koff(tri, ks, P.nl) + (lx - TT_KS * ks)) * nd;
                    for (int ch = 0; ch < nch; ++ch) v += src[(size_t)ch * nst * nd];
                }
                if (have && cX == b && lx < P.nl) {                       // x as a column index l: the sub-strips that reach its block
                    const int lb = lx / TT_LB, ns = (P.nk + R.ksub - 1) / R.ksub, f = tt_dj_first_sub(tri, lb, R.ksub);
                    const double *src = vec + (size_t)(P.dj_l + tt_dj_loff(tri, lb, P.nk, R.ksub) + (lx - TT_LB * lb)) * nd;
                    for (int s2 = f; s2 < ns; ++s2) v += src[(size_t)(s2 - f) * TT_LB * nd];
                }
                if (cX == cj && lx <= lj) v += (lx == lj ? 0.5 : 1.0) * Ee[lx] * xii;   // edge: D[j][l] += w (ij|il) P[i][i]
                acc[u] += v;
            }
        }
        tr_combine(sPart, acc, w, lane, tot);
        if (w == 0) {
#pragma unroll
            for (int u = 0; u < TT_RED_CT; ++u) {
                const int lx = t0 + 64 * u + lane;
                if (lx < nX) R.Dj[d * R.sO + (size_t)jI * N + xs + lx] = tot[u];
            }
        }
    }
}

// kind 0 of a wide pass: ALL densities of the pass in one workgroup.  The DJ vectors are density-minor ([index][ND]: what the tile kernel's
// merge writes with one store per lane), so a lane reads the ND values of its element as one or two 32-byte loads; a workgroup per
// density would fetch every 64-byte sector ND times (measured: 1.0 ms of reduction per density at ND = 8 against 0.5 ms at ND = 1).
template <int ND>
__device__ __forceinline__ void tr_dj_block_wide(const TRArgs &R, const TPairI *__restrict__ pairs, const TRunI *__restrict__ runs,
                                                 const int *__restrict__ jlist, int jI, int cX, int w, int lane, double *sPart)
{
    static_assert(ND == 4 || ND == 8, "densities of a wide pass");
    const int N = R.N;
    const size_t nn = (size_t)N * N;
    const int cj = R.clsI[jI];
    const int xs = R.tab[TVT_CSTART + cX], nX = R.tab[TVT_CSIZE + cX];
    const int lj = jI - R.tab[TVT_CSTART + cj];
    auto add = [](double (&v)[ND], const double *__restrict__ src) {
#pragma unroll
        for (int h = 0; h < ND / 4; ++h) {
            const tt_v4d t = *reinterpret_cast<const tt_v4d *>(src + 4 * h);
            v[4 * h] += t.x; v[4 * h + 1] += t.y; v[4 * h + 2] += t.z; v[4 * h + 3] += t.w;
        }
    };
    for (int t0 = 0; t0 < nX; t0 += 64) {
        const int lx = t0 + lane;
        double acc[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) acc[d] = 0.0;
        for (int q = R.jlist_ptr[jI] + w; q < R.jlist_ptr[jI + 1]; q += TT_RED_WAVES) {
            const int iI = jlist[q];
            const int c = R.clsI[iI] ^ cj;
            const int p = R.tab[TVT_PID + cX * 4 + (cX ^ c)];
            const TPairI P = pairs[(size_t)iI * 10 + p];
            const TRunI Rn = runs[(size_t)iI * 4 + cj];
            const double *vec = R.DJ + (Rn.dj_base + (long long)(jI - Rn.j0) * Rn.dj_len) * ND;
            const int a = R.tab[TVT_PA + p], b = R.tab[TVT_PB + p];
            const bool tri = a == b, have = P.first_task >= 0;
            const long long l0 = Rn.j0 - R.tab[TVT_CSTART + cj];
            const double *Ee = R.T + R.edge_base + Rn.e_base + (long long)lj * (lj + 1) / 2 - l0 * (l0 + 1) / 2;
            double v[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) v[d] = 0.0;
            if (have && cX == a && lx < P.nk) {                           // x as a row index k: the chunks of its stored strip
                const int ks = lx / TT_KS, nst = min(TT_KS, P.nk - TT_KS * ks);
                int nch, wv;
                tt_chunks(tt_nlb(tri, ks, P.nk, P.nl), &nch, &wv);
                const double *src = vec + (size_t)(P.dj_k + tt_dj_koff(tri, ks, P.nl) + (lx - TT_KS * ks)) * ND;
                for (int ch = 0; ch < nch; ++ch) add(v, src + (size_t)ch * nst * ND);
            }
            if (have && cX == b && lx < P.nl) {                           // x as a column index l: the sub-strips that reach its block
                const int lb = lx / TT_LB, ns = (P.nk + R.ksub - 1) / R.ksub, f = tt_dj_first_sub(tri, lb, R.ksub);
                const double *src = vec + (size_t)(P.dj_l + tt_dj_loff(tri, lb, P.nk, R.ksub) + (lx - TT_LB * lb)) * ND;
                for (int s2 = f; s2 < ns; ++s2) add(v, src + (size_t)(s2 - f) * TT_LB * ND);
            }
            if (cX == cj && lx <= lj) {                                   // edge: D[j][l] += w (ij|il) P[i][i]
                const double e = (lx == lj ? 0.5 : 1.0) * Ee[lx];
#pragma unroll
                for (int d = 0; d < ND; ++d) v[d] += e * R.X[(size_t)d * nn + (size_t)iI * N + iI];
            }
#pragma unroll
            for (int d = 0; d < ND; ++d) acc[d] += v[d];
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) sPart[((size_t)w * ND + d) * 64 + lane] = acc[d];
        __syncthreads();
        for (int d = w; d < ND; d += TT_RED_WAVES) {                      // wave d adds the partial rows of density d in fixed order
            double t = 0.0;
            for (int q2 = 0; q2 < TT_RED_WAVES; ++q2) t += sPart[((size_t)q2 * ND + d) * 64 + lane];
            if (lx < nX) R.Dj[d * R.sO + (size_t)jI * N + xs + lx] = t;
        }
        __syncthreads();
    }
}

// kind 1, one workgroup per row k of a class pair: the Jt totals of the row = the partial blocks of every first index above (k, l); the
// waves share the first indices.
__device__ __forceinline__ void tr_jt_block(const TRArgs &R, const TPairI *__restrict__ pairs, int d, int rowid, int w, int lane, double *sPart)
{
    int p = 0, row = rowid;
    while (p + 1 < R.npair && row >= R.tab[TVT_CSIZE + R.tab[TVT_PA + p]]) { row -= R.tab[TVT_CSIZE + R.tab[TVT_PA + p]]; ++p; }
    const int a = R.tab[TVT_PA + p], b = R.tab[TVT_PB + p];
    const bool tri = a == b;
    const int nb = R.tab[TVT_CSIZE + b], pitch = R.tab[TVT_PMPITCH + p];
    const int ncols = tri ? row + 1 : nb;
    const double *Jt = R.Jt + d * R.sJt;
    for (int t0 = 0; t0 < pitch; t0 += 64 * TT_RED_CT) {
        double acc[TT_RED_CT], tot[TT_RED_CT];
#pragma unroll
        for (int u = 0; u < TT_RED_CT; ++u) acc[u] = 0.0;
        if (t0 < ncols)
            for (int iI = w; iI < R.N; iI += TT_RED_WAVES) {
                const TPairI P = pairs[(size_t)iI * 10 + p];
                if (P.first_task < 0 || row >= P.nk) continue;
                const double *src = Jt + P.jt_base + (long long)row * P.jt_pitch;
                const int lim = tri ? row + 1 : P.nl;
                for (int part = 0; part < P.nparts; ++part) {
#pragma unroll
                    for (int u = 0; u < TT_RED_CT; ++u) {
                        const int lc = t0 + 64 * u + lane;
                        if (lc < lim) acc[u] += src[lc];
                    }
                    src += P.jt_part_stride;
                }
            }
        tr_combine(sPart, acc, w, lane, tot);
        if (w == 0) {
#pragma unroll
            for (int u = 0; u < TT_RED_CT; ++u) {
                const int lc = t0 + 64 * u + lane;
                if (lc < pitch) R.JtTot[(size_t)d * R.pm_len + R.tab[TVT_PMOFF + p] + (size_t)row * pitch + lc] = tot[u];
            }
        }
    }
}

// kind 2, one workgroup per i: D[i][x] and Jd[i][j] from the per-task outputs of the tasks of i.  Wave w takes every fourth task (creation
// order) and adds into its own copy of the two output rows in LDS; the copies are added in fixed order.
__device__ __forceinline__ void tr_gather_block(const TRArgs &R, const TTask *__restrict__ tasks, int d, int iI, double *sAcc)
{
    const int N = R.N;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    double *aD = sAcc + (size_t)w * 2 * N, *aJ = aD + N;
    for (int x = lane; x < 2 * N; x += 64) aD[x] = 0.0;
    for (int q = R.itask_ptr[iI] + w; q < R.itask_ptr[iI + 1]; q += TT_RED_WAVES) {
        const TTask *t = tasks + R.itasks[q];
        const int nw = t->nw, nks = t->nks, nj = t->nj, di = t->di_base, jd = t->jd_base;
        if (lane < nks) {                                                   // D[i][k]
            double v = 0.0;
            for (int u = 0; u < nw; ++u) v += R.DIk[d * R.sDIk + (size_t)(di + u) * 64 + lane];
            aD[t->kbase + t->k0 + lane] += v;
        }
        {                                                                   // D[i][l]: lane = (wave of the task, column of its block)
            const int u = lane >> 4, lc = TT_LB * (t->lb0 + u) + (lane & 15);
            if (u < nw && lc < t->ncol) aD[t->lbase + lc] += R.DIl[d * R.sDIl + (size_t)(di + u) * 16 + (lane & 15)];
        }
        if (lane < nj) {                                                    // Jd[i][j]
            double v = 0.0;
            for (int u = 0; u < nw; ++u) v += R.Jd[((size_t)jd + (size_t)u * nj + lane) * R.nd + d];
            aJ[t->j0 + lane] += v;
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < N; x += TT_RED_THREADS) {
        double dd = 0.0, jj = 0.0;
        for (int u = 0; u < TT_RED_WAVES; ++u) { dd += sAcc[(size_t)u * 2 * N + x]; jj += sAcc[(size_t)u * 2 * N + N + x]; }
        R.Di[d * R.sO + (size_t)iI * N + x] = dd;
        R.JD[d * R.sO + (size_t)iI * N + x] = jj;
    }
}

// NDW = 1: one density after the other (block index = density x kinds); NDW = 4 / 8 (a wide pass): the D[j][.] blocks take all densities
// together and come first (they run longest), the gather and Jt blocks follow density by density.
template <int NDW>
__global__ __launch_bounds__(TT_RED_THREADS) void jk_tile_reduce_kernel(const TTask *__restrict__ tasks, const TPairI *__restrict__ pairs,
                                                                         const TRunI *__restrict__ runs, const int *__restrict__ jlist, TRArgs R)
{
    extern __shared__ double sAcc[];                                       // [waves][D row N | J row N] (gather) / [waves][64 TT_RED_CT] (the others)
    const int N = R.N;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n2 = N, n0 = 4 * N, n1 = R.jt_rows;                          // gather first: its workgroups run longest
    int b = blockIdx.x;
    if constexpr (NDW == 1) {
        const int d = b / (n0 + n1 + n2);
        b -= d * (n0 + n1 + n2);
        if (b < n2)
            tr_gather_block(R, tasks, d, N - 1 - b, sAcc);                 // (the last first indices have the most tasks)
        else if (b < n2 + n0)
            tr_dj_block(R, pairs, runs, jlist, d, (b - n2) >> 2, (b - n2) & 3, w, lane, sAcc);
        else
            tr_jt_block(R, pairs, d, b - n2 - n0, w, lane, sAcc);
    } else {
        if (b < n0) { tr_dj_block_wide<NDW>(R, pairs, runs, jlist, b >> 2, b & 3, w, lane, sAcc); return; }
        b -= n0;
        const int d = b / (n1 + n2);
        b -= d * (n1 + n2);
        if (b < n2)
            tr_gather_block(R, tasks, d, N - 1 - b, sAcc);
        else
            tr_jt_block(R, pairs, d, b - n2, w, lane, sAcc);
    }
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
