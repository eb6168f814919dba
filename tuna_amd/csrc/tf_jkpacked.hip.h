// tf_jkpacked.hip.h -- Fock build from the 8-fold symmetry-unique tensor ("packed" layout).
// Reference: calculate_coulomb_matrix tuna_scf.py:55-72 ("ijkl,kl->ij"), calculate_exchange_matrix tuna_scf.py:27-44
// ("ilkj,kl->ij"); the reference keeps all 8 images of every (ij|kl) (pyx:1335-1342), here each unique value is stored once.
//
// Layout: a pair (k >= l) has the padded index tri_off(k) + l, where every row k of the triangle starts at a multiple of
// TF_TRI_PAD = 16 doubles = one 128-byte cache line (tri_off(k) = sum of the row lengths 1, 2, 3, ... each rounded up to 16); pad
// slots hold 0.  Row (i >= j) of the tensor holds (ij|kl) for every pair (k,l) <= (i,j) at [tri_off(k) + l], its length rounded up
// to 16; a rank stores the rows it owns in ascending (i,j) with a row-offset table.  ~1.05 N^4 bytes at N = 400 instead of the
// reference's 8 N^4.  Why whole cache lines: a wave reads 128 columns (1 KB) of a triangle row per load; with rows packed to even
// indices only, that KB straddled 9 lines and the line shared with the neighbouring 128-column chunk -- another workgroup, usually
// on another XCD with its own L2 -- came from HBM twice.  Aligned: 4.7 % more stored bytes, 9 % fewer fetched, kernel 6 % faster.
//
// One pass over row (i,j) has to feed six outputs per element m = (ij|kl):
//     Jd[ij] += m Pp[kl]                      (Pp[kl] = P[k][l] + P[l][k], or P[k][k])
//     Jt[kl] += m Pp[ij]                      (the transposed image (kl|ij); not for kl == ij)
//     D[i][k] += m P[j][l]   D[i][l] += m P[j][k] (k != l)     D[j][k] += m P[i][l] (i != j)    D[j][l] += m P[i][k] (i != j, k != l)
// (the four D terms at half weight when kl == ij), and K = D + D^T covers the transposed images when P is symmetric, as every SCF
// density is.  A general P takes two passes: K = D(P^T) + D(P)^T (the einsum of scf:42 exactly).
//
// Kernel shape: a task = (group of up to JBB rows (i; j0..j0+nr-1) sharing i) x (one chunk of 128 columns); one wave per task, a
// lane owns two adjacent columns (one 16-byte load per row and triangle row k) and walks k up to i.  Tasks are dispatched longest
// first, so the triangle balances itself.  Lane-local accumulators: the "column" sums (outputs indexed by l); the "row" sums
// (outputs indexed by k) are reduced across the wave with a transposing butterfly on permlane swaps / DPP (no LDS) and written
// per task.  Jt partials are written once per group and column (1/JBB of the tensor's bytes) and summed by jk_reduce_kernel.
// No atomics anywhere: results are bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>

#define TF_JKP_JBB 8
#define TF_JKP_CW 128            // columns per chunk (2 per lane)
#define TF_JKP_SEG 16            // segments of the group list in the Jt reduction

// padded triangle: first index of row k, and the stored length of tensor row (i,j)
#ifndef TF_TRI_PAD
#define TF_TRI_PAD 16            // triangle rows and tensor rows start at multiples of this many doubles (a power of two >= 2)
#endif
__host__ __device__ inline long long tri_off(long long k)
{
    const long long q = k / TF_TRI_PAD, r = k % TF_TRI_PAD;     // sum over m = 1..k of m rounded up to the pad
    return TF_TRI_PAD * (TF_TRI_PAD * q * (q + 1) / 2 + r * (q + 1));
}
__host__ __device__ inline long long packed_row_len(long long i, long long j) { return (tri_off(i) + j + TF_TRI_PAD) & ~(long long)(TF_TRI_PAD - 1); }

struct JKGroup {
    int i, j0, nr, r0;           // rows r0..r0+nr-1 (local numbering) = pairs (i, j0..j0+nr-1)
    int roff[TF_JKP_JBB];        // start of row r relative to row 0, in doubles
};
#define TF_JKP_W 4                // groups (waves) per workgroup: their Jt partials are merged in LDS before they are written
#define TF_JKP_KB 4               // triangle rows per merge block
static_assert(TF_JKP_KB <= TF_JKP_W, "one wave per merged row");
// up to TF_JKP_W adjacent groups with the same i share one Jt partial (length = stored length of the first, longest group)
struct JKSuper { int g0, ng; long long yoff, ylen; };
struct JKTask { int super, chunk; };

__device__ __forceinline__ double ld_stream(const double *p) { return __builtin_nontemporal_load(p); }

// Buffer addressing for the streaming loop: a wave-uniform descriptor (base in SGPRs), a wave-uniform byte offset (SGPR) and the
// lane's 32-bit byte offset (one VGPR) -- no per-lane 64-bit address arithmetic.  AUX 2 = non-temporal (touched once).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ double2 buf_load2(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, unsigned uniform_off)
{
    return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lane_off, (int)uniform_off, AUX));
}
template <int AUX>
__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, unsigned uniform_off, double2 v)
{
    typedef unsigned u4_t __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, v), rs, (int)lane_off, (int)uniform_off, AUX);
}

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

// X = P or P^T (dense [N][N], what the exchange terms contract with); Pp[tri_off(k)+l] = P[k][l] + P[l][k] (k != l), P[k][k];
// the pad slots of Pp stay zero (set once at allocation).
__global__ void pack_density_kernel(const double *__restrict__ P, int N, int transpose, double *__restrict__ X, double *__restrict__ Pp)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int k = e / N, l = e - k * N;
    const double a = P[e], b = P[(size_t)l * N + k];
    X[e] = transpose ? b : a;
    if (l <= k) Pp[tri_off(k) + l] = (k == l) ? a : a + b;
}

// The kernel handles ND = 1 or 2 densities per pass.  Its eight "virtual rows" v = d * RB + r are RB = 8 / ND tensor rows times
// ND densities: the loads of a tensor row are shared by the densities, all per-row state is indexed by v.  Arrays of the second
// density follow those of the first at the strides given in JKWave.

// Wave-uniform description of a task (lives in SGPRs).
struct JKWave {
    const double *T0, *P, *Pp;           // first row of the group; density matrices [ND][N][N]; packed densities [ND][NP]
    long long NP;
    int N, i, j0, nr, c0;                // c0 = first column of the chunk
    unsigned roff8[TF_JKP_JBB];          // byte offset of row r from row 0
    double *yg, *DIr_w, *DJr_w;          // density d: + d * ystride / dstrideI / dstrideJ
    size_t rowW, ystride, dstrideI, dstrideJ;
    double ppij[TF_JKP_JBB];             // Pp_d[(i, j_r)] by virtual row
};

// Per-lane state: the two columns l0, l0 + 1 of the lane
template <int ND>
struct JKLane {
    int l0;
    double2 pil[ND], pjl[TF_JKP_JBB];    // P_d[i][l];  P_d[j_r][l] by virtual row
    double2 colI[ND], colJ[TF_JKP_JBB];  // D_d[i][l], D_d[j_r][l] accumulators
};

enum { JKP_DIAG = 1, JKP_FULL = 2 };

// The values a lane needs from triangle row k < i: the tensor elements of its two columns and Pp_d[kl].
template <int ND>
struct JKLoad { double2 m[TF_JKP_JBB / ND], pp[ND]; };

// MODE FULL: every lane has l < k (no masks); DIAG: the 128-column tile on the diagonal (a pair is present iff l0 <= k; the pad
// slot after an odd-length row reads 0).  ALLR: the group has all RB rows.
template <int ND, bool ALLR, int MODE>
__device__ __forceinline__ void jkp_load(JKLoad<ND> &L, int l0, int lane, const JKWave &U, int k)
{
    constexpr int RB = TF_JKP_JBB / ND;
    const long long bk = tri_off(k) + U.c0;
    const bool v = (MODE == JKP_FULL) || l0 <= k;
    const __amdgpu_buffer_rsrc_t rt = buf_rsrc(U.T0 + bk);
    const double2 zero = make_double2(0.0, 0.0);
#pragma unroll
    for (int r = 0; r < RB; ++r) L.m[r] = (v && (ALLR || r < U.nr)) ? buf_load2<2>(rt, 16u * (unsigned)lane, U.roff8[r]) : zero;
#pragma unroll
    for (int d = 0; d < ND; ++d) L.pp[d] = v ? buf_load2<0>(buf_rsrc(U.Pp + d * U.NP + bk), 16u * (unsigned)lane, 0u) : zero;
}

// part 1: everything that needs only the lane's own P values (Jd, Jt, the row sums); part 2: the column sums, which need the
// wave-uniform P[j_r][k], P[i][k] -- fetched through the scalar unit while part 1 runs.
template <int ND>
__device__ __forceinline__ void jkp_row1(const JKLane<ND> &C, const JKLoad<ND> &L, const JKWave &U, double (&jd)[TF_JKP_JBB],
                                         double (&rJ)[TF_JKP_JBB], double (&rI)[ND], double2 (&jt)[ND])
{
    constexpr int RB = TF_JKP_JBB / ND;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        jt[d] = make_double2(0.0, 0.0);
        rI[d] = 0.0;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int v = d * RB + r;
            const double2 m = L.m[r];
            jd[v] += m.x * L.pp[d].x + m.y * L.pp[d].y;
            jt[d].x += m.x * U.ppij[v]; jt[d].y += m.y * U.ppij[v];
            const double2 pj = C.pjl[v];
            rI[d] += m.x * pj.x + m.y * pj.y;
            rJ[v] = m.x * C.pil[d].x + m.y * C.pil[d].y;
        }
    }                                                       // masked lanes loaded zeros: their jt is 0
}

template <int ND, int MODE>
__device__ __forceinline__ void jkp_row2(JKLane<ND> &C, const JKLoad<ND> &L, int k, const double (&pjk)[TF_JKP_JBB], const double (&pik)[ND])
{
    constexpr int RB = TF_JKP_JBB / ND;
    const double o0 = (MODE == JKP_FULL || C.l0 < k) ? 1.0 : 0.0, o1 = (MODE == JKP_FULL || C.l0 + 1 < k) ? 1.0 : 0.0;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const double mx = (MODE == JKP_FULL) ? L.m[r].x : L.m[r].x * o0, my = (MODE == JKP_FULL) ? L.m[r].y : L.m[r].y * o1;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int v = d * RB + r;
            C.colI[d].x += mx * pjk[v]; C.colI[d].y += my * pjk[v];
            C.colJ[v].x += mx * pik[d]; C.colJ[v].y += my * pik[d];
        }
    }
}

// The last triangle row k == i: row r ends at l == j_r, where the element (ij|ij) counts half in K and not at all in Jt.
template <int ND>
__device__ __forceinline__ void jkp_last(JKLane<ND> &C, const JKWave &U, const double (&pjk)[TF_JKP_JBB], const double (&pik)[ND],
                                         double (&jd)[TF_JKP_JBB], double (&rJ)[TF_JKP_JBB], double (&rI)[ND], double2 (&jt2)[ND])
{
    constexpr int RB = TF_JKP_JBB / ND;
    const int i = U.i, jlast = U.j0 + U.nr - 1;
    const long long bk = tri_off(i);
#pragma unroll
    for (int d = 0; d < ND; ++d) { rI[d] = 0.0; jt2[d] = make_double2(0.0, 0.0); }
#pragma unroll
    for (int v = 0; v < TF_JKP_JBB; ++v) rJ[v] = 0.0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int l = C.l0 + e;
        const bool valid = l <= jlast;
        double mrow[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) mrow[r] = (r < U.nr && l <= U.j0 + r) ? ld_stream(U.T0 + (U.roff8[r] >> 3) + bk + l) : 0.0;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double pp = valid ? U.Pp[d * U.NP + bk + l] : 0.0;
            const double pil = e ? C.pil[d].y : C.pil[d].x;
            double jt = 0.0, cI = 0.0;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int v = d * RB + r;
                const double mr = mrow[r];
                const bool diag = (l == U.j0 + r);
                jd[v] += mr * pp;
                jt += diag ? 0.0 : mr * U.ppij[v];
                const double mk = diag ? 0.5 * mr : mr;
                rI[d] += mk * (e ? C.pjl[v].y : C.pjl[v].x);
                rJ[v] += mk * pil;
                const double mc = (l < i) ? mk : 0.0;
                cI += mc * pjk[v];
                if (e) C.colJ[v].y += mc * pik[d]; else C.colJ[v].x += mc * pik[d];
            }
            if (e) { C.colI[d].y += cI; jt2[d].y = jt; } else { C.colI[d].x += cI; jt2[d].x = jt; }   // jt: 0 beyond the group's last column
        }
    }
}

template <int ND, bool ALLR>
__device__ __forceinline__ void jkp_row_sums(const JKWave &U, int k, int lane, double (&rJ)[TF_JKP_JBB], const double (&rI)[ND])
{
    constexpr int RB = TF_JKP_JBB / ND;
    const double tJ = wave_sum8(rJ);
    const int v = lane >> 3, d = v / RB, r = v - d * RB;
    if ((lane & 7) == 0 && (ALLR || r < U.nr)) U.DJr_w[d * U.dstrideJ + (size_t)r * U.rowW + k] = tJ;
#pragma unroll
    for (int dd = 0; dd < ND; ++dd) {
        const double tI = wave_sum1(rI[dd]);
        if (lane == 0) U.DIr_w[dd * U.dstrideI + k] = tI;
    }
}

// Jt of the rows kb..kb+KB-1: the waves of the workgroup have left their partials in slots[kk][wave][d][lane]; wave w adds up
// row kb + w and writes it (fixed order: bitwise reproducible).  Two barriers per block.
template <int ND, int MODE>
__device__ __forceinline__ void jkp_merge_jt(const JKWave &U, double2 *slots, int ng, int w, int lane, int kb, int k1)
{
    __syncthreads();
    const int k = kb + w;
    if (w < TF_JKP_KB && k < k1) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            double2 t = slots[((w * TF_JKP_W) * ND + d) * 64 + lane];
            for (int u = 1; u < ng; ++u) { const double2 x = slots[((w * TF_JKP_W + u) * ND + d) * 64 + lane]; t.x += x.x; t.y += x.y; }
            if (MODE == JKP_FULL || U.c0 + 2 * lane <= k)
                buf_store2<2>(buf_rsrc(U.yg + d * U.ystride + tri_off(k) + U.c0), 16u * (unsigned)lane, 0u, t);
        }
    }
    __syncthreads();
}

template <int ND, bool ALLR>
__device__ __forceinline__ void jkp_uniform_p(const JKWave &U, int k, double (&pjk)[TF_JKP_JBB], double (&pik)[ND])
{
    constexpr int RB = TF_JKP_JBB / ND;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const double *Pd = U.P + (size_t)d * U.N * U.N;
#pragma unroll
        for (int r = 0; r < RB; ++r) pjk[d * RB + r] = (ALLR || r < U.nr) ? Pd[(size_t)(U.j0 + r) * U.N + k] : 0.0;
        pik[d] = Pd[(size_t)U.i * U.N + k];
    }
}

// Rows k0 <= k < k1 of a task in one mode.  The loads of row k + 1 are issued before row k is consumed, so a wave always has a
// full row of requests in flight; the wave-uniform P[j_r][k], P[i][k] come through the scalar unit while part 1 runs.
// Every wave of the workgroup runs the same k range (idle waves included): the barriers of the Jt merge must match.
template <int ND, bool ALLR, int MODE>
__device__ __forceinline__ void jkp_segment(const JKWave &U, JKLane<ND> &C, double (&jd)[TF_JKP_JBB], int k0, int k1, int lane, bool active,
                                            double2 *slots, int ng, int w)
{
    constexpr int JBB = TF_JKP_JBB;
    if (k0 >= k1) return;
    JKLoad<ND> L, Nx;
    if (active) jkp_load<ND, ALLR, MODE>(L, C.l0, lane, U, k0);
    for (int kb = k0; kb < k1; kb += TF_JKP_KB) {
        if (active) {
            const int ke = min(kb + TF_JKP_KB, k1);
            for (int k = kb; k < ke; ++k) {
                jkp_load<ND, ALLR, MODE>(Nx, C.l0, lane, U, min(k + 1, k1 - 1));
                double pjk[JBB], pik[ND], rJ[JBB], rI[ND];
                double2 jt[ND];
                jkp_uniform_p<ND, ALLR>(U, k, pjk, pik);
                jkp_row1<ND>(C, L, U, jd, rJ, rI, jt);
#pragma unroll
                for (int d = 0; d < ND; ++d) slots[(((k - kb) * TF_JKP_W + w) * ND + d) * 64 + lane] = jt[d];
                jkp_row_sums<ND, ALLR>(U, k, lane, rJ, rI);
                jkp_row2<ND, MODE>(C, L, k, pjk, pik);
                L = Nx;
            }
        }
        jkp_merge_jt<ND, MODE>(U, slots, ng, w, lane, kb, k1);
    }
}

template <int ND, bool ALLR>
__device__ __forceinline__ void jkp_task(const JKWave &U, int NW, int c, int lane, int group, int r0, bool active, double2 *slots, int ng,
                                         int w, double *__restrict__ Jd, size_t strideJd, double *__restrict__ DIc, size_t strideDIc,
                                         double *__restrict__ DJc, size_t strideDJc)
{
    constexpr int JBB = TF_JKP_JBB, RB = JBB / ND;
    const int N = U.N, i = U.i;
    JKLane<ND> C;
    C.l0 = U.c0 + 2 * lane;
    {
        const bool in0 = C.l0 < N, in1 = C.l0 + 1 < N;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double *Pd = U.P + (size_t)d * N * N;
            const double *Pi = Pd + (size_t)i * N;
            C.pil[d] = make_double2(in0 ? Pi[C.l0] : 0.0, in1 ? Pi[C.l0 + 1] : 0.0);
            C.colI[d] = make_double2(0.0, 0.0);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const bool have = ALLR || r < U.nr;
                const double *Pj = Pd + (size_t)(U.j0 + r) * N;
                C.pjl[d * RB + r] = make_double2((in0 && have) ? Pj[C.l0] : 0.0, (in1 && have) ? Pj[C.l0 + 1] : 0.0);
                C.colJ[d * RB + r] = make_double2(0.0, 0.0);
            }
        }
    }
    double jd[JBB];
#pragma unroll
    for (int v = 0; v < JBB; ++v) jd[v] = 0.0;

    const int kd1 = min(U.c0 + TF_JKP_CW, i);                         // end of the diagonal tile (exclusive), rows k < i only
    jkp_segment<ND, ALLR, JKP_DIAG>(U, C, jd, U.c0, kd1, lane, active, slots, ng, w);
    jkp_segment<ND, ALLR, JKP_FULL>(U, C, jd, kd1, i, lane, active, slots, ng, w);
    {   // k == i: the groups end at different columns; the first group of the workgroup is the longest
        double2 jt[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) jt[d] = make_double2(0.0, 0.0);
        if (active) {
            double pjk[JBB], pik[ND], rJ[JBB], rI[ND];
            jkp_uniform_p<ND, ALLR>(U, i, pjk, pik);
            jkp_last<ND>(C, U, pjk, pik, jd, rJ, rI, jt);
            jkp_row_sums<ND, ALLR>(U, i, lane, rJ, rI);
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) slots[(w * ND + d) * 64 + lane] = jt[d];
        __syncthreads();
        if (w == 0) {                                  // wave 0 holds the first group: its last column bounds the row; the partial is
                                                       // written (with zeros) up to its padded length, which jk_reduce_kernel sums over
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                double2 t = slots[d * 64 + lane];
                for (int u = 1; u < ng; ++u) { const double2 x = slots[(u * ND + d) * 64 + lane]; t.x += x.x; t.y += x.y; }
                if (C.l0 <= ((U.j0 + U.nr - 1) | (TF_TRI_PAD - 1))) buf_store2<2>(buf_rsrc(U.yg + d * U.ystride + tri_off(i) + U.c0), 16u * (unsigned)lane, 0u, t);
            }
        }
    }
    if (!active) return;
    // column parts (every column l < N of the group belongs to exactly one task)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int l = C.l0 + e;
        if (l < N) {
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                DIc[d * strideDIc + (size_t)group * N + l] = e ? C.colI[d].y : C.colI[d].x;
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    if (ALLR || r < U.nr) DJc[d * strideDJc + (size_t)(r0 + r) * N + l] = e ? C.colJ[d * RB + r].y : C.colJ[d * RB + r].x;
            }
        }
    }
    {
        const double t = wave_sum8(jd);
        const int v = lane >> 3, d = v / RB, r = v - d * RB;
        if ((lane & 7) == 0 && (ALLR || r < U.nr)) Jd[d * strideJd + (size_t)(r0 + r) * NW + c] = t;
    }
}

// Strides (in doubles) between the arrays of density 0 and density 1 of a two-density pass
struct JKStrides { size_t P, Pp, y, Jd, DIc, DIr, DJc, DJr; };

// One workgroup per task (super-group, chunk): wave w owns group g0 + w (idle if the super-group has fewer).  NW = ceil(N / 128)
// chunks; only tasks with 128 chunk <= i exist.  ND densities per pass: groups of 8 / ND rows.
// Outputs: Jd [n_rows][NW] per-task partials; ypart: Jt partials per super-group; DIc [G][N], DJc [n_rows][N]: column parts
// (l-indexed); DIr [G][NW][N], DJr [n_rows][NW][N]: row parts per task (k-indexed, written for 128 chunk <= k <= i).
template <int ND>
__global__ __launch_bounds__(64 * TF_JKP_W) void jk_packed_kernel(const double *__restrict__ T, const long long *__restrict__ rowoff,
                                                                  const JKGroup *__restrict__ groups, const JKSuper *__restrict__ supers,
                                                                  const JKTask *__restrict__ tasks, int N, int NW,
                                                                  const double *__restrict__ P, const double *__restrict__ Pp,
                                                                  double *__restrict__ Jd, double *__restrict__ ypart,
                                                                  double *__restrict__ DIc, double *__restrict__ DIr,
                                                                  double *__restrict__ DJc, double *__restrict__ DJr, JKStrides S)
{
    constexpr int JBB = TF_JKP_JBB, RB = JBB / ND;
    __shared__ double2 slots[TF_JKP_KB * TF_JKP_W * ND * 64];
    const JKTask t = tasks[blockIdx.x];
    const JKSuper sg = supers[t.super];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const bool active = w < sg.ng;
    const int gi = sg.g0 + (active ? w : 0);
    const JKGroup g = groups[gi];
    JKWave U;
    U.T0 = T + rowoff[g.r0]; U.P = P; U.Pp = Pp; U.NP = (long long)S.Pp;
    U.N = N; U.i = g.i; U.j0 = g.j0; U.nr = g.nr; U.c0 = t.chunk * TF_JKP_CW;
    U.yg = ypart + sg.yoff; U.ystride = S.y;
    U.DIr_w = DIr + ((size_t)gi * NW + t.chunk) * N; U.dstrideI = S.DIr;
    U.rowW = (size_t)NW * N;
    U.DJr_w = DJr + ((size_t)g.r0 * NW + t.chunk) * N; U.dstrideJ = S.DJr;
#pragma unroll
    for (int r = 0; r < JBB; ++r) U.roff8[r] = 8u * (unsigned)g.roff[r < RB ? r : 0];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < RB; ++r) U.ppij[d * RB + r] = (r < g.nr) ? Pp[d * S.Pp + tri_off(g.i) + g.j0 + r] : 0.0;
    if (g.nr == RB)
        jkp_task<ND, true>(U, NW, t.chunk, lane, gi, g.r0, active, slots, sg.ng, w, Jd, S.Jd, DIc, S.DIc, DJc, S.DJc);
    else
        jkp_task<ND, false>(U, NW, t.chunk, lane, gi, g.r0, active, slots, sg.ng, w, Jd, S.Jd, DIc, S.DIc, DJc, S.DJc);
}

// Jt partial sums over the padded pair index q: super-groups are sorted by descending partial length, so those that cover q are
// a prefix of the list.  Block (bx, by) of a (ceil(NP/256), nseg) grid: segment by sums its slice of that prefix; out[by][q].
__device__ __forceinline__ void jt_reduce_block(int bx, int by, int nseg, const double *__restrict__ ypart, const JKSuper *__restrict__ groups,
                                                int n_groups, long long NP, double *__restrict__ out)
{
    const long long q = (long long)bx * 256 + threadIdx.x;
    if (q >= NP) return;
    const int per = (n_groups + nseg - 1) / nseg;
    const int g0 = by * per, g1 = min(n_groups, g0 + per);
    // the partials that cover every column of this block (a prefix: found by a wave-uniform bisection) are summed without the
    // per-element test, so their loads can be issued ahead; the few that end inside the block follow with the test
    const long long qmax = min(NP - 1, (long long)bx * 256 + 255);
    int lo = g0, hi = g1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (groups[mid].ylen > qmax) lo = mid + 1; else hi = mid;
    }
    double s = 0.0;
#pragma unroll 8
    for (int g = g0; g < lo; ++g) s += ypart[groups[g].yoff + q];
    for (int g = lo; g < g1; ++g) {
        if (groups[g].ylen <= q) break;
        s += ypart[groups[g].yoff + q];
    }
    out[(size_t)by * NP + q] = s;
}

// D[a][x] = sum over groups with i == a of (column part + row parts of the chunks)[x] + the same over owned rows (i > a, j == a).
// Block (a, bx) of an (N, ceil(N/64)) grid, 256 threads = 4 slices x 64 columns; gfirst[a]..gfirst[N+a] are the groups with i == a.
__device__ __forceinline__ void kd_reduce_block(int a, int bx, double *sPart, const double *__restrict__ DIc, const double *__restrict__ DIr,
                                                const double *__restrict__ DJc, const double *__restrict__ DJr, int NW,
                                                const int *__restrict__ gfirst, const int *__restrict__ rowmap, int N, double *__restrict__ D)
{
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int x = bx * 64 + lane;
    const int nw = min(NW, x / TF_JKP_CW + 1);            // chunks that start at or before x
    double s = 0.0;
    if (x < N) {
        if (x <= a)
            for (int g0 = gfirst[a] + sl, ge = gfirst[N + a]; g0 < ge; g0 += 16) {
                double t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = (g0 + 4 * u < ge) ? DIc[(size_t)(g0 + 4 * u) * N + x] : 0.0;
                for (int w = 0; w < nw; ++w) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (g0 + 4 * u < ge) t[u] += DIr[((size_t)(g0 + 4 * u) * NW + w) * N + x];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (g0 + 4 * u < ge) s += t[u];
            }
        // four rows of the slice at a time: their row lookups, then all their loads, are issued together (the loads depend on the
        // lookups; one row at a time left a single dependent chain per lane); summed in row order as before
        for (int i0 = max(a + 1, x) + sl; i0 < N; i0 += 16) {
            int r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + 4 * u; r[u] = (i < N) ? rowmap[(size_t)i * (i + 1) / 2 + a] : -1; }
            double t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = (r[u] >= 0) ? DJc[(size_t)r[u] * N + x] : 0.0;
            for (int w = 0; w < nw; ++w) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (r[u] >= 0) t[u] += DJr[((size_t)r[u] * NW + w) * N + x];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r[u] >= 0) s += t[u];
        }
    }
    sPart[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && x < N) D[(size_t)a * N + x] = ((sPart[lane] + sPart[64 + lane]) + sPart[128 + lane]) + sPart[192 + lane];
}

// Both reductions of all densities of a pass in ONE launch (they are independent and each alone leaves most of the chip idle):
// per density first the N * ceil(N/64) exchange blocks, then the ceil(NP/256) * nseg transposed-Coulomb blocks.  Fixed summation
// order inside every block: bitwise reproducible.
struct JKReduce {
    const double *ypart, *DIc, *DIr, *DJc, *DJr;
    double *Jt, *D[2];
    const JKSuper *supers;
    const int *gfirst, *rowmap;
    long long NP;
    size_t sy, sJt, sDIc, sDIr, sDJc, sDJr;              // strides between densities
    int n_supers, nseg, N, NW;
};
__global__ __launch_bounds__(256) void jk_reduce_kernel(JKReduce R)
{
    __shared__ double sPart[256];
    const int gxK = (R.N + 63) / 64, nK = R.N * gxK;
    const int gxJ = (int)((R.NP + 255) / 256), nJ = gxJ * R.nseg;
    int b = blockIdx.x;
    const int d = b / (nK + nJ);
    b -= d * (nK + nJ);
    if (b < nK)
        kd_reduce_block(b / gxK, b % gxK, sPart, R.DIc + d * R.sDIc, R.DIr + d * R.sDIr, R.DJc + d * R.sDJc, R.DJr + d * R.sDJr, R.NW, R.gfirst,
                        R.rowmap, R.N, R.D[d]);
    else {
        b -= nK;
        jt_reduce_block(b % gxJ, b / gxJ, R.nseg, R.ypart + d * R.sy, R.supers, R.n_supers, R.NP, R.Jt + d * R.sJt);
    }
}

// K = D + D2^T (D2 = D for a symmetric density; for a general one D = D(P^T), D2 = D(P));
// J[a][b] = sum over the chunks of Jd[row(ab)] (owned rows) + sum_s Jt_s[pair(ab)]
__global__ void jk_packed_final_kernel(const double *__restrict__ D, const double *__restrict__ D2, const double *__restrict__ Jd, int NW,
                                       const double *__restrict__ Jt, int nseg, const int *__restrict__ rowmap, int N,
                                       double *__restrict__ J, double *__restrict__ K)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int a = e / N, b = e - a * N;
    K[e] = D[e] + D2[(size_t)b * N + a];
    const int hi = max(a, b), lo = min(a, b);
    const long long NP = tri_off(N), q = tri_off(hi) + lo;
    const int r = rowmap[(size_t)hi * (hi + 1) / 2 + lo];
    double s = 0.0;
    if (r >= 0)
        for (int w = 0; w <= hi / TF_JKP_CW; ++w) s += Jd[(size_t)r * NW + w];
    for (int t = 0; t < nseg; ++t) s += Jt[(size_t)t * NP + q];
    J[e] = s;
}

struct OutRowP {
    int i, j;              // output AO indices (i >= j)
    int cartA, cartB;      // first Cartesian AO of the two bra shells
    int ncb, pad;
    long long slab_off;    // first slab row of this bra pair
    long long dst_off;     // offset of the packed row in the stored tensor
};

// packed tensor row (i,j) = bra transform of the ket-transformed slab, keeping only pairs (k >= l) up to (i,j); pad slots <- 0.
// grid (ceil(N / 4), rows): a block takes four triangle rows k (one per wave), lanes run over l -- the padded index is
// tri_off(k) + l, no index inversion per element; the (<= 6 x 6) Cartesian -> spherical terms of the two bra AOs are staged in LDS
// once per block (nested sums in the order of xform_bra_store: the same rounding).
#define TF_XBP_KR 4
__global__ __launch_bounds__(64 * TF_XBP_KR) void xform_bra_store_packed(const double *__restrict__ in, double *__restrict__ eri,
                                                                       const OutRowP *__restrict__ rows, long long row_len, int ld,
                                                                       const int *__restrict__ ptr, const int *__restrict__ idx,
                                                                       const double *__restrict__ val)
{
    __shared__ double sValA[32], sValB[32];
    __shared__ long long sOffA[32], sOffB[32];
    const OutRowP R = rows[blockIdx.y];
    const int k0 = TF_XBP_KR * blockIdx.x;
    if (k0 > R.i) return;
    const int pa = ptr[R.i], na = min(32, ptr[R.i + 1] - pa), pb = ptr[R.j], nb = min(32, ptr[R.j + 1] - pb);
    if (threadIdx.x < na) { sValA[threadIdx.x] = val[pa + threadIdx.x]; sOffA[threadIdx.x] = (long long)(idx[pa + threadIdx.x] - R.cartA) * R.ncb * row_len; }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + nb) {
        const int t = threadIdx.x - 64;
        sValB[t] = val[pb + t]; sOffB[t] = (long long)(idx[pb + t] - R.cartB) * row_len;
    }
    __syncthreads();
    const int k = k0 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k > R.i) return;
    const long long tk = tri_off(k);
    // stored slots of triangle row k in this tensor row: the whole padded row for k < i, up to the row's padded end for k == i
    const int lenk = (k < R.i) ? (int)(tri_off(k + 1) - tk) : (int)(packed_row_len(R.i, R.j) - tk);
    const int lmax = (k < R.i) ? k : R.j;                           // pairs (k, l <= lmax) exist; beyond: pad slots
    const double *__restrict__ src = in + R.slab_off * row_len + (long long)k * ld;
    double *__restrict__ dst = eri + R.dst_off + tk;
    for (int l = lane; l < lenk; l += 64) {
        double s = 0.0;
        if (l <= lmax) {
            for (int qa = 0; qa < na; ++qa) {
                double t = 0.0;
                for (int qb = 0; qb < nb; ++qb) t += sValB[qb] * src[sOffA[qa] + sOffB[qb] + l];
                s += sValA[qa] * t;
            }
        }
        dst[l] = s;
    }
}

__device__ __forceinline__ double packed_element(const double *__restrict__ eri, const int *__restrict__ rowmap,
                                                 const long long *__restrict__ rowoff, int i, int j, int k, int l)
{
    const int ih = max(i, j), il = min(i, j), kh = max(k, l), kl = min(k, l);
    const long long p = (long long)ih * (ih + 1) / 2 + il, q = (long long)kh * (kh + 1) / 2 + kl;
    const int r = rowmap[max(p, q)];
    if (r < 0) return 0.0;
    return eri[rowoff[r] + (p >= q ? tri_off(kh) + kl : tri_off(ih) + il)];
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
