// tf_jkpacked.hip.h -- Fock build from the symmetry-unique, parity-blocked tensor ("packed" layout).
// Reference: calculate_coulomb_matrix tuna_scf.py:55-72 ("ijkl,kl->ij"), calculate_exchange_matrix tuna_scf.py:27-44
// ("ilkj,kl->ij"); the reference keeps all 8 images of every (ij|kl) (pyx:1335-1342) and the exact zeros of the x/y reflection
// parity rule (pyx:1324-1327).  Here each unique value is stored once and the parity zeros are not stored at all.
//
// Parity classes.  On a z-axis diatomic every AO (Cartesian or real spherical) is even or odd under x -> -x and under y -> -y:
// class(AO) = (x parity) | (y parity) << 1.  (ij|kl) vanishes unless class(i) ^ class(j) == class(k) ^ class(l).
//
// Layout (tests/layout_model.py is a NumPy model of everything below and is checked against the reference einsums on a CPU):
//   * internal AO index sigma(k) = cstart[class(k)] + loc(k): the AOs sorted by class (larger classes first), original order
//     inside a class; loc(k) = rank of k in its class.  Densities and the exchange partial sums live in internal indices.
//   * tensor row (i >= j) (original order), class c = class(i) ^ class(j), stores for every AO k <= i ONE segment: the values
//     (ij|kl) for the AOs l <= k of class class(k) ^ c in ascending l -- cnt[c][k] of them, padded to TF_SEG_PAD doubles with
//     zeros (in the segment of k == i the slots beyond l == j hold zeros too).  The segments of a row are ordered by
//     (class of k, k): four sections, so that a task -- one class of k against one class of l -- streams contiguous memory.
//     All rows with the same (i, c) have the same shape.  ~N^4 / 4 bytes (+ padding) instead of the reference's 8 N^4.
//   * rows are stored in UNITS of up to 8 rows (i; j0..j0+nr-1) -- consecutive j of one class, the row groups of the kernel -- with
//     their segments interleaved: unit base + nr (secoff[a] + offA[k]) + p padcnt(k) + loc(l) for the row at position p.  What a
//     wave reads in one step (the segment of k of its 8 rows) is then ONE contiguous run of memory (DRAM pages are opened for
//     ~3 KB instead of ~400 bytes per row).
//   * pair index of (k >= l), class c: cbase[c] + fullsec[c][class(k)] + offA[c][sigma(k)] + loc(l) -- the shape of a complete
//     row; the packed density Pp and the transposed-Coulomb partials use it.
//
// One pass over row (i,j) has to feed six outputs per element m = (ij|kl):
//     Jd[ij] += m Pp[kl]                      (Pp[kl] = P[k][l] + P[l][k], or P[k][k])
//     Jt[kl] += m Pp[ij]                      (the transposed image (kl|ij); not for kl == ij)
//     D[i][k] += m P[j][l]   D[i][l] += m P[j][k] (k != l)     D[j][k] += m P[i][l] (i != j)    D[j][l] += m P[i][k] (i != j, k != l)
// (the four D terms at half weight when kl == ij), and K = D + D^T covers the transposed images when P is symmetric, as every SCF
// density is.  A general P takes two passes: K = D(P^T) + D(P)^T (the einsum of scf:42 exactly).
//
// Kernel shape: a task = (up to TF_JKP_W groups of up to JBB rows (i; j0..j0+nr-1) sharing i and the class of j) x (one chunk of
// <= 128 columns l of one class b); one wave per group, a lane owns two adjacent columns (one 16-byte load per row and k) and walks
// the AOs k <= i of class a = b ^ c whose segment reaches the chunk.  Lane-local accumulators: the "column" sums (outputs indexed
// by l); the "row" sums (outputs indexed by k) are reduced across the wave with a transposing butterfly on permlane swaps / DPP
// (no LDS) and written per task.  Jt partials are merged over the waves of a workgroup in LDS, written once per super-group and
// summed by jk_reduce_kernel.  No atomics anywhere: results are bitwise reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include "tf_layout.hip.h"

#define TF_JKP_JBB 8               // rows of a storage unit (and the largest row group)
#ifndef TF_JKP_VR1
#define TF_JKP_VR1 8               // rows of a group in a one-density pass: 8 (251 VGPRs, 2 waves per SIMD) or 4 (151 VGPRs, 3 waves per
                                   // SIMD, twice the steps: measured the same 2.1 ms at N = 400)
#endif
#ifndef TF_JKP_OCC4
#define TF_JKP_OCC4 3              // waves per SIMD the 4-row shape is compiled for (4: 128 VGPRs with 41 spilled, 2.4 ms)
#endif
template <int ND> struct JKShape {           // virtual rows v = d * RB + r of a pass: RB tensor rows times ND densities
    static constexpr int VR = ND == 1 ? TF_JKP_VR1 : 8, RB = VR / ND;
};
#ifndef TF_JKP_GPW
#define TF_JKP_GPW 2               // row groups a wave works on at once: 2 (half waves on 64 columns) or 4 (quarter waves on 32 columns:
                                   // 20 % fewer wave steps at N = 400, but measured 8-15 % SLOWER -- DESIGN.md section 4.1)
#endif
#define TF_JKP_LG (64 / TF_JKP_GPW)   // lanes per group
#define TF_JKP_CW (2 * TF_JKP_LG)  // columns per chunk (2 per lane of a group's lanes)
#define TF_JKP_SEG 16            // segments of the super-group lists in the Jt reduction
#ifndef TF_JKP_STAGES
#define TF_JKP_STAGES 2           // register buffers of the load ring: the loads of STAGES - 1 steps are in flight (2 or 4)
#endif


// Weight proxy of the shard plan (tf_shard_plan_pairs works from shell dimensions alone): the row lengths of the unblocked
// triangle, every triangle row rounded up to TF_TRI_PAD.  The blocked rows are ~1/4 of that, uniformly enough for balancing.
#ifndef TF_TRI_PAD
#define TF_TRI_PAD 16
#endif
__host__ __device__ inline long long tri_off(long long k)
{
    const long long q = k / TF_TRI_PAD, r = k % TF_TRI_PAD;     // sum over m = 1..k of m rounded up to the pad
    return TF_TRI_PAD * (TF_TRI_PAD * q * (q + 1) / 2 + r * (q + 1));
}
__host__ __device__ inline long long packed_row_len(long long i, long long j) { return (tri_off(i) + j + TF_TRI_PAD) & ~(long long)(TF_TRI_PAD - 1); }

struct JKGroup {
    int i, j0, nr, r0;           // rows r0..r0+nr-1 (local numbering) = pairs (i, j0..j0+nr-1), internal indices
    int c, lamj0;                // class of the rows; loc of j0
    int unr, p0;                 // rows of the storage unit that holds the group; position of the group's first row in it
    long long ub;                // base of the unit in the tensor
    int secoff[4];               // start of section a inside a row of this group
};
#ifndef TF_JKP_W
#define TF_JKP_W 4                // waves per workgroup (TF_JKP_GPW groups each): their Jt partials are merged in LDS before they are written
#endif
#define TF_JKP_KB 4               // k steps per merge block
// up to 2 TF_JKP_W adjacent groups with the same i and class share one Jt partial (complete-row shape: NP[c] doubles at yoff)
struct JKSuper { int g0, ng, c, i; long long yoff; int ke[4]; };   // ke[a] = cntA[a][i]: the rows reach the members kappa < ke[a] of class a
struct JKTask { int super, w, part, pad; };   // part: which stretch of KS steps of the walk (the walks are cut for several ranks: shorter tasks)

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

// X[sigma(r)][sigma(c)] = P[r][c] (or P[c][r]): what the exchange terms contract with, in internal indices;
// Pp[pair index of (k >= l)] = P[k][l] + P[l][k] (k != l), P[k][k]; the pad slots of Pp stay zero (set once at allocation).
__global__ void pack_density_kernel(const double *__restrict__ P, BLayout L, int transpose, double *__restrict__ X, double *__restrict__ Pp)
{
    const int N = L.N;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int k = e / N, l = e - k * N;
    const double a = P[e], b = P[(size_t)l * N + k];
    const int wk = L.ao[k], wl = L.ao[l];
    const int sk = ao_sigma(L, wk), sl = ao_sigma(L, wl);
    X[(size_t)sk * N + sl] = transpose ? b : a;
    if (l <= k) {
        const int c = ao_cls(wk) ^ ao_cls(wl);
        Pp[bl_cbase(L, c) + bl_fullsec(L, c, ao_cls(wk)) + L.kinfo[c * N + sk].offA + ao_loc(wl)] = (k == l) ? a : a + b;
    }
}

// out[0] = largest |P[k][l]| of all, out[1] = largest |P[k][l]| between AOs of different x/y parity classes (ORIGINAL indices, one
// workgroup per row), as bit patterns of non-negative doubles merged with atomicMax (a NaN counts as a huge cross element).
__global__ void class_cross_max_kernel(const double *__restrict__ P, BLayout L, unsigned long long *__restrict__ out)
{
    __shared__ double sa[256], sx[256];
    const int N = L.N, k = blockIdx.x, ck = ao_cls(L.ao[k]);
    double all = 0.0, cross = 0.0;
    for (int l = threadIdx.x; l < N; l += 256) {
        double v = fabs(P[(size_t)k * N + l]);
        if (v != v) v = 1e300;
        all = fmax(all, v);
        if (ao_cls(L.ao[l]) != ck) cross = fmax(cross, v);
    }
    sa[threadIdx.x] = all; sx[threadIdx.x] = cross;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) { sa[threadIdx.x] = fmax(sa[threadIdx.x], sa[threadIdx.x + st]); sx[threadIdx.x] = fmax(sx[threadIdx.x], sx[threadIdx.x + st]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(out, (unsigned long long)__double_as_longlong(sa[0]));
        if (sx[0] != 0.0) atomicMax(out + 1, (unsigned long long)__double_as_longlong(sx[0]));
    }
}

// The kernel handles ND = 1 or 2 densities per pass.  Its eight "virtual rows" v = d * RB + r are RB = 8 / ND tensor rows times
// ND densities: the loads of a tensor row are shared by the densities, all per-row state is indexed by v.  Arrays of the second
// density follow those of the first at the strides given in JKWave.
//
// A wave works on TWO row groups at once: lanes 0-31 (half 0) on group 2w, lanes 32-63 (half 1) on group 2w + 1 of the super-group,
// both on the same 64 columns (2 per lane) and the same AO k.  The segments of this layout are short (a triangle of classes: on
// average half a class long), so 64-column chunks with 16 rows per wave keep far more lanes busy than 128 columns x 8 rows did
// (N = 400: 1.66 instead of 2.60 million wave steps per build).  Everything that depends on the half lives in VGPRs.

// 32-lane versions of the wave sums: the two halves of a wave reduce independently
__device__ __forceinline__ double pair_step4(double a, double b)       // lanes with bit 2 clear: a[l] + a[l + 4]; set: b[l - 4] + b[l]
{
    double recv = dpp_merge<0x12C, 0x5>(a, a);                          // row_ror:12: banks 0,2 <- a[lane + 4]
    recv = dpp_merge<0x124, 0xA>(recv, b);                              // row_ror:4:  banks 1,3 <- b[lane - 4]
    const double keep = dpp_merge<TF_DPP_QUAD_IDENT, 0xA>(a, b);        // banks 1,3 keep b
    return keep + recv;
}
__device__ __forceinline__ double quad_sum(double t)
{
    t += dpp_merge<TF_DPP_QUAD_XOR1, 0xF>(t, t);
    t += dpp_merge<TF_DPP_QUAD_XOR2, 0xF>(t, t);
    return t;
}
// eight per-lane values -> lane q of a half holds that half's total of value (q >> 2) & 7
__device__ __forceinline__ double half_sum8(const double (&v)[8])
{
    const double w0 = pair_step16(v[0], v[4]), w1 = pair_step16(v[1], v[5]), w2 = pair_step16(v[2], v[6]), w3 = pair_step16(v[3], v[7]);
    const double u0 = pair_step8(w0, w2), u1 = pair_step8(w1, w3);
    return quad_sum(pair_step4(u0, u1));
}
__device__ __forceinline__ double half_sum1(double t)                   // every lane gets the total of its half
{
    t = pair_step16(t, t);
    t += dpp_merge<TF_DPP_ROR8, 0xF>(t, t);
    return sum8(t);
}

// 16-lane versions: the four quarters of a wave reduce independently
// (DPP bank masks select groups of four lanes, not lanes inside a quad: the step across lane bit 1 takes a lane predicate)
__device__ __forceinline__ double pair_step2(double a, double b, bool bit1)   // lanes with bit 1 clear: a[l] + a[l + 2]; set: b[l - 2] + b[l]
{
    const double keep = bit1 ? b : a, give = bit1 ? a : b;
    return keep + dpp_merge<TF_DPP_QUAD_XOR2, 0xF>(give, give);
}
// eight per-lane values -> lane q of a quarter holds that quarter's total of value (q >> 1) & 7
__device__ __forceinline__ double quarter_sum8(const double (&v)[8])
{
    const double w0 = pair_step8(v[0], v[4]), w1 = pair_step8(v[1], v[5]), w2 = pair_step8(v[2], v[6]), w3 = pair_step8(v[3], v[7]);
    const double u0 = pair_step4(w0, w2), u1 = pair_step4(w1, w3);
    double t = pair_step2(u0, u1, (__lane_id() & 2u) != 0u);
    t += dpp_merge<TF_DPP_QUAD_XOR1, 0xF>(t, t);
    return t;
}
__device__ __forceinline__ double quarter_sum1(double t)                // every lane gets the total of its quarter
{
    t += dpp_merge<TF_DPP_ROR8, 0xF>(t, t);
    return sum8(t);
}
// the sums over the lanes of a group; lane q of a group holds value (q >> TF_JKP_VSH) & 7 in the lanes with (q & TF_JKP_VMASK) == 0 (and its copies)
#if TF_JKP_GPW == 4
#define TF_JKP_VSH 1
__device__ __forceinline__ double group_sum8(const double (&v)[8]) { return quarter_sum8(v); }
__device__ __forceinline__ double group_sum1(double t) { return quarter_sum1(t); }
__device__ __forceinline__ double across_groups(double t) { t = pair_step32(t, t); return pair_step16(t, t); }   // same lane of every group
#else
#define TF_JKP_VSH 2
__device__ __forceinline__ double group_sum8(const double (&v)[8]) { return half_sum8(v); }
__device__ __forceinline__ double group_sum1(double t) { return half_sum1(t); }
__device__ __forceinline__ double across_groups(double t) { return pair_step32(t, t); }
#endif
#define TF_JKP_VMASK ((1 << TF_JKP_VSH) - 1)

#define TF_BUF_OOB 0x80000000u       // lane offset beyond num_records (0x7fffffff) of buf_rsrc: loads return 0, stores are dropped
template <int AUX>
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, unsigned uniform_off, double v)
{
    typedef unsigned u2_t __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2_t, v), rs, (int)lane_off, (int)uniform_off, AUX);
}
template <int AUX>
__device__ __forceinline__ double buf_load1(__amdgpu_buffer_rsrc_t rs, unsigned lane_off, unsigned uniform_off)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)lane_off, (int)uniform_off, AUX));
}

// Wave-uniform description of a task (lives in SGPRs).
struct JKWave {
    const double *Tb, *X, *Pp;           // the lowest storage unit of the wave's groups; densities [ND][N][N] (internal); packed densities (class
                                         // base + section a) [ND][NPtot]
    const KInfo *kinfo;                  // kinfo[c] + cstart[a]: indexed by kappa
    long long NPtot;
    int N, i;                            // AOs; internal first index of the rows
    int kI0, c0, lam0, width, cm;        // first internal AO of the k class; first internal column, its loc, columns of the chunk;
                                         // cm = 1 when k and l are of the same class (the pair (k,k) exists)
    int sec;                             // start of section a in a row of these groups (same i and class: same shape)
    __amdgpu_buffer_rsrc_t rt;           // the tensor from the lowest storage unit of the wave's groups + the chunk's first column
    int rp;                              // row parts of this chunk inside a row part vector
    double *yg, *DIr, *DJr;              // Jt partial; row parts of the workgroup's first group / row (density d: + d * ystride / dstrideI / dstrideJ)
    size_t RS, ystride, dstrideI, dstrideJ;
};

// Per-lane state: the half and its group, the two columns of the lane (loc lam, lam + 1 of class b; internal index lI, lI + 1)
template <int ND>
struct JKLane {
    int h, q, lam, lI;
    int nr, lamj0, r0, g;                // rows of this half's group (0: none), loc of its first j, its first local row, its index
    unsigned xrow;                       // byte offset of X[j0][0]
    unsigned gb, unr, p0;                // the group's storage unit: base (doubles from U.Tb), rows, position of the group in it
    unsigned djoff, dioff;               // byte offsets of the group's row parts of this chunk from U.DJr / U.DIr
    double ppij[JKShape<ND>::VR];             // Pp_d[(i, j_r)] by virtual row
    double2 pil[ND], pjl[JKShape<ND>::VR];    // P_d[i][l];  P_d[j_r][l] by virtual row
    double2 colI[ND], colJ[JKShape<ND>::VR];  // D_d[i][l], D_d[j_r][l] accumulators
};

enum { JKP_DIAG = 1, JKP_FULL = 2 };

// The values a lane needs from the segment of AO k: the tensor elements of its two columns and Pp_d[kl]; cnt = stored values.
template <int ND>
struct JKLoad { double2 m[JKShape<ND>::RB], pp[ND]; int cnt; };

// MODE FULL: every lane has l < k (no masks); DIAG: a pair is present iff lam < cnt (the slot after an odd count reads 0).
// ALLR: both halves have a group with all RB rows.
template <int ND, bool ALLR, int MODE>
__device__ __forceinline__ void jkp_load(JKLoad<ND> &L, const JKLane<ND> &C, const JKWave &U, int kap)
{
    constexpr int RB = JKShape<ND>::RB;
    const KInfo ki = U.kinfo[kap];
    const int pc = (ki.cnt + TF_SEG_PAD - 1) & ~(TF_SEG_PAD - 1);          // padded segment: the rows of a unit follow each other at this stride
    L.cnt = ki.cnt;
    const bool v = (MODE == JKP_FULL) || C.lam < ki.cnt;
    // lane offset of the group's segment of k: unit base + rows of the unit * (section + segment offset) + position * padded length,
    // 24-bit multiply-adds (full rate); the descriptor is the same for the whole task
    const __amdgpu_buffer_rsrc_t rt = U.rt;
    const unsigned seg = __umul24(C.p0, (unsigned)pc) + __umul24(C.unr, (unsigned)(U.sec + ki.offA)) + C.gb;
    const unsigned dsel = 8u * seg;
#ifdef TF_ABL_ALLVALID
    const unsigned voff = 16u * (unsigned)C.q + dsel;      // (timing experiment only: every lane of every row loads)
#define TF_ABL_ROWOK(r) true
#else
    const unsigned voff = v ? 16u * (unsigned)C.q + dsel : TF_BUF_OOB;
#define TF_ABL_ROWOK(r) (ALLR || (r) < C.nr)
#endif
#pragma unroll
#ifdef TF_ABL_NOTENSOR
    for (int r = 0; r < RB; ++r) L.m[r] = make_double2(C.ppij[r] + (double)kap, C.ppij[r] - (double)voff);
#else
    for (int r = 0; r < RB; ++r) L.m[r] = buf_load2<2>(rt, TF_ABL_ROWOK(r) ? voff : TF_BUF_OOB, 8u * (unsigned)(r * pc));
#endif
    const long long bk = (long long)ki.offA + U.lam0;
#pragma unroll
    for (int d = 0; d < ND; ++d) L.pp[d] = buf_load2<0>(buf_rsrc(U.Pp + d * U.NPtot + bk), v ? 16u * (unsigned)C.q : TF_BUF_OOB, 0u);
}

// part 1: everything that needs only the lane's own P values (Jd, Jt, the row sums); part 2: the column sums, which need
// P[j_r][k] (per half: a broadcast vector load issued before part 1) and the wave-uniform P[i][k] (scalar unit).
template <int ND>
__device__ __forceinline__ void jkp_row1(const JKLane<ND> &C, const JKLoad<ND> &L, double (&jd)[JKShape<ND>::VR],
                                         double (&rJ)[JKShape<ND>::VR], double (&rI)[ND], double2 (&jt)[ND])
{
    constexpr int RB = JKShape<ND>::RB;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        jt[d] = make_double2(0.0, 0.0);
        rI[d] = 0.0;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int v = d * RB + r;
            const double2 m = L.m[r];
            jd[v] += m.x * L.pp[d].x + m.y * L.pp[d].y;
            jt[d].x += m.x * C.ppij[v]; jt[d].y += m.y * C.ppij[v];
            const double2 pj = C.pjl[v];
            rI[d] += m.x * pj.x + m.y * pj.y;
            rJ[v] = m.x * C.pil[d].x + m.y * C.pil[d].y;
        }
    }                                                       // masked lanes loaded zeros: their jt is 0
}

template <int ND, int MODE>
__device__ __forceinline__ void jkp_row2(JKLane<ND> &C, const JKLoad<ND> &L, const JKWave &U, const double (&pjk)[JKShape<ND>::VR], const double (&pik)[ND])
{
    constexpr int RB = JKShape<ND>::RB;
    const int lim = L.cnt - U.cm;                           // columns strictly below k (the pair (k,k) has no column term)
    const double o0 = (MODE == JKP_FULL || C.lam < lim) ? 1.0 : 0.0, o1 = (MODE == JKP_FULL || C.lam + 1 < lim) ? 1.0 : 0.0;
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

// P_d[j_r][k] of the lane's half (one 8-byte broadcast load per virtual row) and the wave-uniform P_d[i][k]
template <int ND, bool ALLR>
__device__ __forceinline__ void jkp_p_k(const JKLane<ND> &C, const JKWave &U, int kI, double (&pjk)[JKShape<ND>::VR], double (&pik)[ND])
{
    constexpr int RB = JKShape<ND>::RB;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const double *Pd = U.X + (size_t)d * U.N * U.N;
        const __amdgpu_buffer_rsrc_t rx = buf_rsrc(Pd);
#pragma unroll
#ifdef TF_ABL_NOPJK
        for (int r = 0; r < RB; ++r) pjk[d * RB + r] = C.ppij[d * RB + r];
#else
        for (int r = 0; r < RB; ++r) pjk[d * RB + r] = buf_load1<0>(rx, (ALLR || r < C.nr) ? C.xrow : TF_BUF_OOB, 8u * (unsigned)(r * U.N + kI));
#endif
        pik[d] = Pd[(size_t)U.i * U.N + kI];
    }
}

// The segment of k == i (tasks whose k class is that of i): row r ends at l == j_r, where the element (ij|ij) counts half in K
// and not at all in Jt.
template <int ND>
__device__ __forceinline__ void jkp_last(JKLane<ND> &C, const JKWave &U, const double (&pjk)[JKShape<ND>::VR], const double (&pik)[ND],
                                         double (&jd)[JKShape<ND>::VR], double (&rJ)[JKShape<ND>::VR], double (&rI)[ND], double2 (&jt2)[ND])
{
    constexpr int RB = JKShape<ND>::RB;
    const int i = U.i, lamlast = C.lamj0 + C.nr - 1;
    const KInfo kiL = U.kinfo[i - U.kI0];
    const long long bk = kiL.offA;
    const int pc = (kiL.cnt + TF_SEG_PAD - 1) & ~(TF_SEG_PAD - 1);
    const double *Tk = U.Tb + ((long long)C.gb + (long long)C.unr * (U.sec + kiL.offA) + (long long)C.p0 * pc);   // segment of k == i, first row of the lane's group
#pragma unroll
    for (int d = 0; d < ND; ++d) { rI[d] = 0.0; jt2[d] = make_double2(0.0, 0.0); }
#pragma unroll
    for (int v = 0; v < JKShape<ND>::VR; ++v) rJ[v] = 0.0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int lam = C.lam + e;
        const bool valid = lam <= lamlast;
        double mrow[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) mrow[r] = (r < C.nr && lam <= C.lamj0 + r) ? ld_stream(Tk + r * pc + lam) : 0.0;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double pp = valid ? U.Pp[d * U.NPtot + bk + lam] : 0.0;
            const double pil = e ? C.pil[d].y : C.pil[d].x;
            double jt = 0.0, cI = 0.0;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int v = d * RB + r;
                const double mr = mrow[r];
                const bool diag = (lam == C.lamj0 + r);
                jd[v] += mr * pp;
                jt += diag ? 0.0 : mr * C.ppij[v];
                const double mk = diag ? 0.5 * mr : mr;
                rI[d] += mk * (e ? C.pjl[v].y : C.pjl[v].x);
                rJ[v] += mk * pil;
                const double mc = (C.lI + e != i) ? mk : 0.0;
                cI += mc * pjk[v];
                if (e) C.colJ[v].y += mc * pik[d]; else C.colJ[v].x += mc * pik[d];
            }
            if (e) { C.colI[d].y += cI; jt2[d].y = jt; } else { C.colI[d].x += cI; jt2[d].x = jt; }   // jt: 0 beyond the group's last column
        }
    }
}

template <int ND, bool ALLR>
__device__ __forceinline__ void jkp_row_sums(const JKLane<ND> &C, const JKWave &U, int kap, double (&rJ)[JKShape<ND>::VR], const double (&rI)[ND])
{
    constexpr int VR = JKShape<ND>::VR, RB = JKShape<ND>::RB;
    const int v = (C.q >> TF_JKP_VSH) & 7;
    const bool first = (C.q & TF_JKP_VMASK) == 0;
    // (stores without branches as well: lanes that have nothing to store use an out-of-range offset)
    if constexpr (VR == 8) {
        const double tJ = group_sum8(rJ);
        const int d = v / RB, r = v - d * RB;
        buf_store1<0>(buf_rsrc(U.DJr + kap), (first && r < C.nr) ? C.djoff + 8u * (unsigned)(d * U.dstrideJ + (size_t)r * U.RS) : TF_BUF_OOB, 0u, tJ);
#pragma unroll
        for (int dd = 0; dd < ND; ++dd) {
            const double tI = group_sum1(rI[dd]);
            buf_store1<0>(buf_rsrc(U.DIr + dd * U.dstrideI + kap), (C.q == 0 && C.nr > 0) ? C.dioff : TF_BUF_OOB, 0u, tI);
        }
    } else {
        // four rows, one density: the row sums and the sum of the first index share one butterfly (values 0-3: rows, 4: the first index)
        static_assert(ND == 1 && VR == 4, "shapes: 8 virtual rows, or 4 rows of one density");
        const double vals[8] = {rJ[0], rJ[1], rJ[2], rJ[3], rI[0], 0.0, 0.0, 0.0};
        const double t = group_sum8(vals);
        buf_store1<0>(buf_rsrc(U.DJr + kap), (first && v < C.nr) ? C.djoff + 8u * (unsigned)((size_t)v * U.RS) : TF_BUF_OOB, 0u, t);
        buf_store1<0>(buf_rsrc(U.DIr + kap), (first && v == 4 && C.nr > 0) ? C.dioff : TF_BUF_OOB, 0u, t);
    }
}

// workgroup barrier that waits for this wave's LDS traffic only (__syncthreads() also drains the vector memory loads in flight)
__device__ __forceinline__ void jkp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Jt of the steps kb..kb+KB-1: the waves of the workgroup have left their partials in slots[kk][wave][d][lane]; wave w adds up
// step kb + w over the waves and the two halves and writes it (fixed order: bitwise reproducible).  Two barriers per block.
template <int ND, int MODE>
__device__ __forceinline__ void jkp_merge_jt(const JKWave &U, double2 *slots, int nw, int w, int lane, int kb, int k1)
{
    jkp_lds_barrier();
    const int nwg = (int)(blockDim.x >> 6);                          // waves of this workgroup (4, 2 or 1): each merges every nwg-th step
    for (int kk = w; kk < TF_JKP_KB; kk += nwg) {
        const int kap = kb + kk;
        if (kap >= k1) break;
        const KInfo ki = U.kinfo[kap];
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            double2 t = slots[((kk * TF_JKP_W) * ND + d) * 64 + lane];
            for (int u = 1; u < nw; ++u) { const double2 x = slots[((kk * TF_JKP_W + u) * ND + d) * 64 + lane]; t.x += x.x; t.y += x.y; }
            t.x = across_groups(t.x); t.y = across_groups(t.y);                   // the groups of the wave hold the same columns
            if (lane < TF_JKP_LG && (MODE == JKP_FULL || U.lam0 + 2 * lane < ki.cnt))
                buf_store2<2>(buf_rsrc(U.yg + d * U.ystride + ki.offA + U.lam0), 16u * (unsigned)lane, 0u, t);
        }
    }
    jkp_lds_barrier();
}

// Steps k0 <= kappa < k1 of a task in one mode.  The kernel is bound by memory latency, not by arithmetic: what counts is the number
// of bytes a SIMD has in flight.  The loads of the next step are issued before a step is consumed, every step is ONE basic block
// (no branch around a load or a store: see jkp_load), and the barriers of the Jt merge wait for LDS traffic only -- __syncthreads()
// would drain the prefetched loads as well.  Every wave of the workgroup runs the same range (idle waves included): the barriers
// must match.

template <int ND, bool ALLR, int MODE>
__device__ __forceinline__ void jkp_segment(const JKWave &U, JKLane<ND> &C, double (&jd)[JKShape<ND>::VR], int k0, int k1, int lane, bool active,
                                            double2 *slots, int nw, int w)
{
    constexpr int JBB = JKShape<ND>::VR, S = TF_JKP_STAGES;
    static_assert(TF_JKP_KB % S == 0, "the ring position of a step must be a compile-time constant");
    if (k0 >= k1) return;
    // ring of S register buffers: step kk of a merge block uses R[kk % S] and issues the loads of step kk + S - 1 into R[(kk + S - 1) % S].
    // (A plain "L = Nx" copy at the end of a step would make every step wait for the loads it has just issued.)
    JKLoad<ND> R[S];
    if (active) {
#pragma unroll
        for (int s = 0; s < S - 1; ++s) jkp_load<ND, ALLR, MODE>(R[s], C, U, min(k0 + s, k1 - 1));
    }
    for (int kb = k0; kb < k1; kb += TF_JKP_KB) {
        if (active) {
#pragma unroll
            for (int kk = 0; kk < TF_JKP_KB; ++kk) {
                const int k = kb + kk;
                if (k < k1) {
                    double pjk[JBB], pik[ND], rJ[JBB], rI[ND];
                    double2 jt[ND];
                    // P[j_r][k] first: vector memory operations complete in order, and part 2 must not wait for the prefetched tensor loads
                    jkp_p_k<ND, ALLR>(C, U, U.kI0 + k, pjk, pik);
                    jkp_load<ND, ALLR, MODE>(R[(kk + S - 1) % S], C, U, min(k + S - 1, k1 - 1));
#ifdef TF_ABL_LOADSONLY
                    {
#pragma unroll
                        for (int r = 0; r < JBB / ND; ++r) jd[r] += R[kk % S].m[r].x + R[kk % S].m[r].y;
                        jd[0] += R[kk % S].pp[0].x + pjk[0] + pik[0];
                    }
                    continue;
#endif
                    jkp_row1<ND>(C, R[kk % S], jd, rJ, rI, jt);
#ifdef TF_ABL_NOMERGE
                    {
                        const KInfo ki = U.kinfo[k];
#pragma unroll
                        for (int d = 0; d < ND; ++d) {
                            double2 t = jt[d];
                            t.x = across_groups(t.x); t.y = across_groups(t.y);
                            buf_store2<2>(buf_rsrc(U.yg + d * U.ystride + ki.offA + U.lam0), (lane < TF_JKP_LG && (MODE == JKP_FULL || U.lam0 + 2 * lane < ki.cnt)) ? 16u * (unsigned)lane : TF_BUF_OOB, 0u, t);
                        }
                    }
#else
#pragma unroll
                    for (int d = 0; d < ND; ++d) slots[((kk * TF_JKP_W + w) * ND + d) * 64 + lane] = jt[d];
#endif
                    jkp_row_sums<ND, ALLR>(C, U, k, rJ, rI);
                    jkp_row2<ND, MODE>(C, R[kk % S], U, pjk, pik);
                }
            }
        }
#ifndef TF_ABL_NOMERGE
        jkp_merge_jt<ND, MODE>(U, slots, nw, w, lane, kb, k1);
#endif
    }
}

// kap0 <= kappa < kd1: masked steps; kd1 <= kappa < klim: unmasked; then (last) the segment of k == i.
template <int ND, bool ALLR>
__device__ __forceinline__ void jkp_task(const JKWave &U, JKLane<ND> &C, int NW, int wchunk, int lane, bool active, double2 *slots, int nw,
                                         int w, int kap0, int kd1, int klim, bool last, double *__restrict__ Jd, size_t strideJd,
                                         double *__restrict__ DIc, size_t strideDIc, double *__restrict__ DJc, size_t strideDJc)
{
    constexpr int JBB = JKShape<ND>::VR, RB = JBB / ND;
    const int N = U.N, i = U.i;
    const bool in0 = 2 * C.q < U.width, in1 = 2 * C.q + 1 < U.width;
    {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const double *Pd = U.X + (size_t)d * N * N;
            const double *Pi = Pd + (size_t)i * N;
            C.pil[d] = make_double2(in0 ? Pi[C.lI] : 0.0, in1 ? Pi[C.lI + 1] : 0.0);
            C.colI[d] = make_double2(0.0, 0.0);
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const bool have = r < C.nr;
                const double *Pj = Pd + (C.xrow >> 3) + (size_t)r * N;
                C.pjl[d * RB + r] = make_double2((in0 && have) ? Pj[C.lI] : 0.0, (in1 && have) ? Pj[C.lI + 1] : 0.0);
                C.colJ[d * RB + r] = make_double2(0.0, 0.0);
            }
        }
    }
    double jd[JBB];
#pragma unroll
    for (int v = 0; v < JBB; ++v) jd[v] = 0.0;

    jkp_segment<ND, ALLR, JKP_DIAG>(U, C, jd, kap0, kd1, lane, active, slots, nw, w);
    jkp_segment<ND, ALLR, JKP_FULL>(U, C, jd, kd1, klim, lane, active, slots, nw, w);
    if (last) {   // k == i: the rows end at different columns
        double2 jt[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) jt[d] = make_double2(0.0, 0.0);
        if (active) {
            double pjk[JBB], pik[ND], rJ[JBB], rI[ND];
            jkp_p_k<ND, false>(C, U, i, pjk, pik);
            jkp_last<ND>(C, U, pjk, pik, jd, rJ, rI, jt);
            jkp_row_sums<ND, ALLR>(C, U, i - U.kI0, rJ, rI);
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) slots[(w * ND + d) * 64 + lane] = jt[d];
        __syncthreads();
        if (w == 0) {                                  // the whole segment is written (zeros beyond the rows' last columns)
            const KInfo ki = U.kinfo[i - U.kI0];
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                double2 t = slots[d * 64 + lane];
                for (int u = 1; u < nw; ++u) { const double2 x = slots[(u * ND + d) * 64 + lane]; t.x += x.x; t.y += x.y; }
                t.x = across_groups(t.x); t.y = across_groups(t.y);
                if (lane < TF_JKP_LG && U.lam0 + 2 * lane < ki.cnt) buf_store2<2>(buf_rsrc(U.yg + d * U.ystride + ki.offA + U.lam0), 16u * (unsigned)lane, 0u, t);
            }
        }
    }
    if (!active) return;
    // column parts (every column of the chunk belongs to this task alone), per half: each half has its own group
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if ((e ? in1 : in0) && C.nr > 0) {
            const int l = C.lI + e;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                DIc[d * strideDIc + (size_t)C.g * N + l] = e ? C.colI[d].y : C.colI[d].x;
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    if (r < C.nr) DJc[d * strideDJc + (size_t)(C.r0 + r) * N + l] = e ? C.colJ[d * RB + r].y : C.colJ[d * RB + r].x;
            }
        }
    }
    {
        double vals[8];
#pragma unroll
        for (int v = 0; v < 8; ++v) vals[v] = v < JBB ? jd[v < JBB ? v : 0] : 0.0;
        const double t = group_sum8(vals);
        const int v = (C.q >> TF_JKP_VSH) & 7, d = v / RB, r = v - d * RB;
        if ((C.q & TF_JKP_VMASK) == 0 && v < JBB && r < C.nr) Jd[d * strideJd + (size_t)(C.r0 + r) * NW + wchunk] = t;
    }
}

// Strides (in doubles) between the arrays of density 0 and density 1 of a two-density pass
struct JKStrides { size_t P, Pp, y, Jd, DIc, DIr, DJc, DJr;      // between the arrays of density 0 and density 1
                   size_t planeJd, planeI, planeJ; int KS, MP; };  // between the column-part / Jd planes of the parts of a walk; steps per part, parts

// One workgroup per task (super-group, chunk): wave w owns the groups g0 + 2w (half 0) and g0 + 2w + 1 (half 1), idle if the
// super-group has fewer.  Only tasks with at least one step exist (kap0[c][w] < cntA[a][i]).  ND densities per pass: groups of
// 8 / ND rows.
// Outputs: Jd [n_rows][NW] per-task partials; ypart: Jt partials per super-group; DIc [G][N], DJc [n_rows][N]: column parts
// (l-indexed); DIr [G][RS], DJr [n_rows][RS]: row parts per task (k-indexed: entry rpoff[c][w] + kappa, written for every step).
template <int ND>
__global__ __launch_bounds__(64 * TF_JKP_W, TF_JKP_STAGES > 2 ? 1 : (JKShape<ND>::VR == 4 ? TF_JKP_OCC4 : 2)) void jk_packed_kernel(const double *__restrict__ T, const JKGroup *__restrict__ groups,
                                                                  const JKSuper *__restrict__ supers,
                                                                  const JKTask *__restrict__ tasks, BLayout L,
                                                                  const KInfo *__restrict__ kinfo /* = L.kinfo: a __restrict__ kernel
                                                                  argument is read through the scalar unit, a pointer inside L is not */,
                                                                  const double *__restrict__ X, const double *__restrict__ Pp,
                                                                  double *__restrict__ Jd, double *__restrict__ ypart,
                                                                  double *__restrict__ DIc, double *__restrict__ DIr,
                                                                  double *__restrict__ DJc, double *__restrict__ DJr, JKStrides S)
{
    constexpr int JBB = JKShape<ND>::VR, RB = JBB / ND;
    __shared__ double2 slots[TF_JKP_KB * TF_JKP_W * ND * 64];
    const JKTask t = tasks[blockIdx.x];
    const JKSuper sg = supers[t.super];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int GPW = TF_JKP_GPW;
    const int nw = (sg.ng + GPW - 1) / GPW;                            // waves with at least one group
    const bool active = w < nw;
    // the wave's groups, the one that lies lowest in memory first (the lane offsets of the others are unsigned): a lower unit, or -- groups
    // that are parts of one storage unit -- the lower position in the unit.  Missing groups repeat group 0 with no rows.
    int ids[GPW], ngw = 0;
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
        const int gi = sg.g0 + (active ? GPW * w : 0) + g;
        const bool have = active && GPW * w + g < sg.ng;
        ids[g] = have ? gi : sg.g0 + (active ? GPW * w : 0);
        ngw += have ? 1 : 0;
    }
    {
        auto below = [&](int x, int y) {                               // group x lies below group y
            const long long ux = groups[x].ub, uy = groups[y].ub;
            return ux < uy || (ux == uy && groups[x].p0 < groups[y].p0);
        };
        // selection of the lowest into ids[0] is all the kernel needs; the present groups stay in front of the repeats
        int lowest = 0;
#pragma unroll
        for (int g = 1; g < GPW; ++g) if (g < ngw && below(ids[g], ids[lowest])) lowest = g;
        const int tswap = ids[0]; ids[0] = ids[lowest]; ids[lowest] = tswap;
#pragma unroll
        for (int g = 0; g < GPW; ++g) ids[g] = __builtin_amdgcn_readfirstlane(ids[g]);   // (wave-uniform: keeps the group records in SGPRs)
    }
    const JKGroup gA = groups[ids[0]];
    const int N = L.N, NW = L.NW, c = sg.c;
    const int b = L.chunk_cls[t.w], a = b ^ c;
    JKWave U;
    U.N = N; U.i = __builtin_amdgcn_readfirstlane(gA.i);
    U.kI0 = bl_cstart(L, a); U.c0 = L.chunk_c0[t.w]; U.lam0 = U.c0 - bl_cstart(L, b); U.width = L.chunk_width[t.w]; U.cm = (c == 0) ? 1 : 0;
    U.sec = groups[ids[0]].secoff[a];                                   // (table read from memory: a is a run-time index)
    U.Tb = T + gA.ub; U.X = X; U.NPtot = (long long)S.Pp;
    U.rt = buf_rsrc(U.Tb + U.lam0);
    const long long pbase = bl_cbase(L, c) + bl_fullsec(L, c, a);
    U.Pp = Pp + pbase;
    U.kinfo = kinfo + (size_t)c * N + U.kI0;
    U.yg = ypart + sg.yoff + bl_fullsec(L, c, a); U.ystride = S.y;
    U.RS = (size_t)L.RS;
    U.rp = L.rpoff[c * NW + t.w];
    JKLane<ND> C;
    C.h = lane / TF_JKP_LG; C.q = lane % TF_JKP_LG;
    C.lam = U.lam0 + 2 * C.q; C.lI = U.c0 + 2 * C.q;
    // per-lane copy of the lane's group; row parts: uniform bases at the lowest group / row of the wave, small per-lane offsets
    int gmin = ids[0], rmin = gA.r0;
    C.nr = active ? gA.nr : 0; C.lamj0 = gA.lamj0; C.r0 = gA.r0; C.g = ids[0];
    C.gb = 0u; C.unr = (unsigned)gA.unr; C.p0 = (unsigned)gA.p0;
    int j0 = gA.j0;
    bool all_full = active && gA.nr == RB;
#pragma unroll
    for (int g = 1; g < GPW; ++g) {
        const JKGroup gG = groups[ids[g]];
        const bool have = g < ngw;
        gmin = min(gmin, ids[g]); rmin = min(rmin, gG.r0);
        all_full = all_full && have && gG.nr == RB;
        if (C.h == g) {
            C.nr = have ? gG.nr : 0; C.lamj0 = gG.lamj0; C.r0 = gG.r0; C.g = ids[g]; j0 = gG.j0;
            C.gb = (unsigned)(gG.ub - gA.ub); C.unr = (unsigned)gG.unr; C.p0 = (unsigned)gG.p0;
        }
    }
    U.DIr = DIr + (size_t)gmin * L.RS + U.rp; U.dstrideI = S.DIr;
    U.DJr = DJr + (size_t)rmin * L.RS + U.rp; U.dstrideJ = S.DJr;
    C.xrow = 8u * (unsigned)(j0 * N);
    C.dioff = 8u * (unsigned)((C.g - gmin) * L.RS);
    C.djoff = 8u * (unsigned)((C.r0 - rmin) * L.RS);
    {
        // Pp[(i, j_r)]: pair index of (i, j_r) -- i is the larger original index of the row
        const int ci = L.clsI[gA.i];
        const long long pij = bl_cbase(L, c) + bl_fullsec(L, c, ci) + kinfo[(size_t)c * N + gA.i].offA + C.lamj0;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int r = 0; r < RB; ++r) C.ppij[d * RB + r] = (r < C.nr) ? Pp[d * S.Pp + pij + r] : 0.0;
    }
    // the walk kap0 <= kappa < ke of (super-group, chunk), part t.part of it: KS steps; the segment of k == i belongs to the last part
    const int kap00 = L.kap0[c * NW + t.w], ke = supers[t.super].ke[a];
    const int kap0 = kap00 + t.part * S.KS, kend = min(ke, kap0 + S.KS);
    const bool last = (a == L.clsI[gA.i]) && kend == ke;
    const int klim = kend - (last ? 1 : 0);
    const int kd1 = min(max(L.kapF[c * NW + t.w], kap0), klim);
    Jd += (size_t)t.part * S.planeJd; DIc += (size_t)t.part * S.planeI; DJc += (size_t)t.part * S.planeJ;   // column parts and Jd: one plane per part
    if (all_full)
        jkp_task<ND, true>(U, C, NW, t.w, lane, active, slots, nw, w, kap0, kd1, klim, last, Jd, S.Jd, DIc, S.DIc, DJc, S.DJc);
    else
        jkp_task<ND, false>(U, C, NW, t.w, lane, active, slots, nw, w, kap0, kd1, klim, last, Jd, S.Jd, DIc, S.DIc, DJc, S.DJc);
}

// Jt partial sums over the pair index q of class c: the super-groups of a class are sorted by descending original i, so those whose
// rows reach the AO k of q are a prefix of the class's list (ke[a] is non-increasing along it).  Block (bx, by) of a
// (sum_c ceil(NP[c] / 256), nseg) grid: segment by sums its slice of every class's list; out[by][q].
#ifndef TF_JKR_THREADS
#define TF_JKR_THREADS 256         // (measured: 1024-thread blocks, 16 slices, are slower: 0.41 against 0.34 ms of tail at N = 400) threads of a jk_reduce_kernel block: 64 lanes x TF_JKR_THREADS / 64 slices of the rows / groups of an index
#endif
struct JKJtPlan { int sfirst[5]; int bfirst[5]; };   // supers of class c: sfirst[c] .. sfirst[c + 1]; blocks of class c: bfirst[c] .. bfirst[c + 1]
__device__ __forceinline__ void jt_reduce_block(int bx, int by, int nseg, const double *__restrict__ ypart, const JKSuper *__restrict__ supers,
                                                const JKJtPlan &JP, const BLayout &L, double *__restrict__ out)
{
    int c = 0;
    while (c < 3 && bx >= JP.bfirst[c + 1]) ++c;
    const long long q = (long long)(bx - JP.bfirst[c]) * TF_JKR_THREADS + threadIdx.x;
    if (q >= bl_np(L, c)) return;
    const int kI = L.gk[bl_gbase(L, c) + (int)(q / TF_SEG_PAD)];
    const int a = L.clsI[kI], kap = kI - bl_cstart(L, a);
    const KInfo ki = L.kinfo[(size_t)c * L.N + kI];
    double s = 0.0;
    if ((int)(q - bl_fullsec(L, c, a)) - ki.offA < ki.cnt) {                    // (pad slots: 0)
        const int n = JP.sfirst[c + 1] - JP.sfirst[c], per = (n + nseg - 1) / nseg;
        const int s0 = JP.sfirst[c] + by * per, s1 = min(JP.sfirst[c + 1], s0 + per);
        for (int g = s0; g < s1; ++g) {
            if (supers[g].ke[a] <= kap) break;
            s += ypart[supers[g].yoff + q];
        }
    }
    out[(size_t)by * L.NPtot + bl_cbase(L, c) + q] = s;
}

// D[x][y] (internal indices) = sum over the groups whose rows have first index x of (column part + row parts)[y]
//                            + the same over the owned rows (z, x) with second index x (jrows[jptr[x] .. jptr[x + 1])).
// The row parts of a group / row are a dense [MC][N] block (MC = most chunks of one class): the task of chunk number s of its class
// writes slot s at the internal index of k, so the parts that belong to output column y are simply column y of every slot.  No
// validity tests: the set of entries a pass writes does not depend on the density, the buffers are zeroed once at allocation, and
// entries no task writes stay zero.  Block (x, bx) of an (N, ceil(N/128)) grid, 256 threads = 4 slices x 64 lanes x 2 columns;
// four partial vectors at a time so that their loads are in flight together; fixed summation order: bitwise reproducible.
// (every load of a partial vector -- the column part and up to TF_JKR_MCMAX row-part slots -- is issued before the first add: with a
// run-time slot loop the four rounds of a vector were four memory latencies one after the other, and that chain, not bandwidth, was the
// 0.30 ms of this kernel at N = 400)
#ifndef TF_JKR_U
#define TF_JKR_U 2                 // partial vectors of a slice in flight together (measured at N = 400: 1: 0.247 ms, 2: 0.251, 3: 0.270, 4: 0.271, 8: 0.322)
#endif
#ifndef TF_JKR_MCMAX
#define TF_JKR_MCMAX 4
#endif
struct KdLoads { double2 c, s[TF_JKR_MCMAX]; };
__device__ __forceinline__ void kd_issue(KdLoads &q, const double *__restrict__ colp, const double *__restrict__ rowp, int N, int MC, int y, bool two, bool on)
{
    const double2 z = make_double2(0.0, 0.0);
    q.c = !on ? z : (two ? *reinterpret_cast<const double2 *>(colp + y) : make_double2(colp[y], 0.0));
#pragma unroll
    for (int sl = 0; sl < TF_JKR_MCMAX; ++sl)
        q.s[sl] = !(on && sl < MC) ? z : (two ? *reinterpret_cast<const double2 *>(rowp + (size_t)sl * N + y) : make_double2(rowp[(size_t)sl * N + y], 0.0));
}
__device__ __forceinline__ double2 kd_sum(const KdLoads &q, const double *__restrict__ rowp, int N, int MC, int y, bool two, bool on)
{
    double2 t = q.c;
#pragma unroll
    for (int sl = 0; sl < TF_JKR_MCMAX; ++sl) { t.x += q.s[sl].x; t.y += q.s[sl].y; }
    for (int sl = TF_JKR_MCMAX; sl < MC && on; ++sl) {                   // (more chunks per class than the unrolled part holds: N > 512)
        const double2 v = two ? *reinterpret_cast<const double2 *>(rowp + (size_t)sl * N + y) : make_double2(rowp[(size_t)sl * N + y], 0.0);
        t.x += v.x; t.y += v.y;
    }
    return t;
}
__device__ __forceinline__ double2 kd_vec(const double *__restrict__ colp, const double *__restrict__ rowp, int N, int MC, int y, bool two)
{
    double2 t = two ? *reinterpret_cast<const double2 *>(colp + y) : make_double2(colp[y], 0.0);
    for (int sl = 0; sl < MC; ++sl) {
        const double2 v = two ? *reinterpret_cast<const double2 *>(rowp + (size_t)sl * N + y) : make_double2(rowp[(size_t)sl * N + y], 0.0);
        t.x += v.x; t.y += v.y;
    }
    return t;
}

__device__ __forceinline__ void kd_reduce_block(int x, int bx, double2 *sPart, const double *__restrict__ DIc, const double *__restrict__ DIr,
                                                const double *__restrict__ DJc, const double *__restrict__ DJr, int N, int MC, int MP,
                                                size_t planeI, size_t planeJ,
                                                const int *__restrict__ gfirst, const int *__restrict__ jptr, const int2 *__restrict__ jrows,
                                                const int *__restrict__ origI, double *__restrict__ D)
{
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int y = bx * 128 + 2 * lane;
    const size_t RS = (size_t)MC * N;
    const bool two = ((N & 1) == 0) && y + 1 < N;           // 16-byte loads need even row strides
    double2 s = make_double2(0.0, 0.0);
    if (y < N) {
        // A row (i, j) holds values (ij|kl) with k <= i, l <= k in ORIGINAL order, so its partial sums at an index above i were never
        // written (they are the zeros of the allocation): those rows / groups are skipped -- the same sums, half of the bytes.
        const int yo = (y + 1 < N) ? min(origI[y], origI[y + 1]) : origI[y];
        const bool groups_reach = origI[x] >= yo;
        for (int pass = 0; pass < (two || y + 1 >= N ? 1 : 2); ++pass) {
            const int yy = y + pass;
            double2 acc = make_double2(0.0, 0.0);
            constexpr int NSL = TF_JKR_THREADS / 64;                   // slices
            for (int g0 = gfirst[x] + sl, ge = groups_reach ? gfirst[N + x] : 0; g0 < ge; g0 += TF_JKR_U * NSL) {
                KdLoads q[TF_JKR_U];
#pragma unroll
                for (int u = 0; u < TF_JKR_U; ++u) {
                    const int g = g0 + NSL * u;
                    kd_issue(q[u], DIc + (size_t)g * N, DIr + (size_t)g * RS, N, MC, yy, two, g < ge);
                }
#pragma unroll
                for (int u = 0; u < TF_JKR_U; ++u) {
                    const int g = g0 + NSL * u;
                    double2 t = kd_sum(q[u], DIr + (size_t)g * RS, N, MC, yy, two, g < ge);
                    for (int pl = 1; pl < MP; ++pl)                 // column parts of the further parts of a cut walk
                        if (g < ge) { const double2 v = kd_vec(DIc + pl * planeI + (size_t)g * N, DIr, N, 0, yy, two); t.x += v.x; t.y += v.y; }
                    acc.x += t.x; acc.y += t.y;
                }
            }
            for (int p0 = jptr[x] + sl, pe = jptr[x + 1]; p0 < pe; p0 += TF_JKR_U * NSL) {
                KdLoads q[TF_JKR_U];
                int rr[TF_JKR_U];
#pragma unroll
                for (int u = 0; u < TF_JKR_U; ++u) {
                    const int p = p0 + NSL * u;
                    const int2 e = (p < pe) ? jrows[p] : make_int2(-1, 0);
                    rr[u] = (e.x >= 0 && e.y >= yo) ? e.x : -1;            // (a row whose first index lies below both columns holds nothing for them)
                }
#pragma unroll
                for (int u = 0; u < TF_JKR_U; ++u)
                    kd_issue(q[u], DJc + (size_t)max(rr[u], 0) * N, DJr + (size_t)max(rr[u], 0) * RS, N, MC, yy, two, rr[u] >= 0);
#pragma unroll
                for (int u = 0; u < TF_JKR_U; ++u) {
                    const int r = rr[u];
                    double2 t = kd_sum(q[u], DJr + (size_t)max(r, 0) * RS, N, MC, yy, two, r >= 0);
                    for (int pl = 1; pl < MP; ++pl)
                        if (r >= 0) { const double2 v = kd_vec(DJc + pl * planeJ + (size_t)r * N, DJr, N, 0, yy, two); t.x += v.x; t.y += v.y; }
                    acc.x += t.x; acc.y += t.y;
                }
            }
            if (pass == 0) s = acc; else s.y = acc.x;
        }
    }
    sPart[threadIdx.x] = s;
    __syncthreads();
    if (sl == 0 && y < N) {
        double2 a = sPart[lane];
        for (int q = 1; q < TF_JKR_THREADS / 64; ++q) { const double2 b = sPart[64 * q + lane]; a.x += b.x; a.y += b.y; }   // fixed order
        D[(size_t)x * N + y] = a.x;
        if (y + 1 < N) D[(size_t)x * N + y + 1] = a.y;
    }
}

// Both reductions of all densities of a pass in ONE launch (they are independent and each alone leaves most of the chip idle):
// per density first the N * ceil(N/64) exchange blocks, then the Jt blocks.  Fixed summation order inside every block: bitwise
// reproducible.
struct JKReduce {
    const double *ypart, *DIc, *DIr, *DJc, *DJr;
    double *Jt, *D[2];
    const JKSuper *supers;
    const int *gfirst, *jptr;
    const int2 *jrows;                                   // (local row, original first index) of the rows listed by second index
    const int *xorder;                                   // output rows x of the exchange blocks, most partial vectors first (dispatch order)
    const int2 *row_ij;                                  // original (i, j) of the local rows
    int MC, MP;                                          // slots of a row part vector: most chunks of one class; parts of a walk
    size_t planeI, planeJ;                               // between the column-part planes of the parts
    JKJtPlan jp;
    size_t sy, sJt, sDIc, sDIr, sDJc, sDJr;              // strides between densities
    int nseg;
    int dbg;                                             // timing experiments (TF_JKR_DBG: 1 = exchange blocks only, 2 = Jt blocks only)
};
__global__ __launch_bounds__(TF_JKR_THREADS) void jk_reduce_kernel(JKReduce R, BLayout L)
{
    __shared__ double2 sPart[TF_JKR_THREADS];
    const int gxK = (L.N + 127) / 128, nK = L.N * gxK;
    const int gxJ = R.jp.bfirst[4], nJ = gxJ * R.nseg;
    int b = blockIdx.x;
    const int d = b / (nK + nJ);
    b -= d * (nK + nJ);
    if (R.dbg && (b < nK) == (R.dbg == 2)) return;
    if (b < nK)
        kd_reduce_block(R.xorder ? R.xorder[b / gxK] : b / gxK, b % gxK, sPart, R.DIc + d * R.sDIc, R.DIr + d * R.sDIr, R.DJc + d * R.sDJc, R.DJr + d * R.sDJr, L.N, R.MC, R.MP,
                        R.planeI, R.planeJ, R.gfirst, R.jptr, R.jrows, L.origI, R.D[d]);
    else {
        b -= nK;
        jt_reduce_block(b % gxJ, b / gxJ, R.nseg, R.ypart + d * R.sy, R.supers, R.jp, L, R.Jt + d * R.sJt);
    }
}

// Original indices (a, b): K = D + D2^T (D2 = D for a symmetric density; for a general one D = D(P^T), D2 = D(P));
// J[a][b] = sum over the chunks with a task of Jd[row(ab)] (owned rows) + sum_s Jt_s[pair(ab)]
__global__ void jk_packed_final_kernel(const double *__restrict__ D, const double *__restrict__ D2, const double *__restrict__ Jd,
                                       size_t planeJd, int KS, int MP, const double *__restrict__ Jt, int nseg, const int *__restrict__ rowmap, BLayout L,
                                       double *__restrict__ J, double *__restrict__ K)
{
    const int N = L.N, NW = L.NW;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int a = e / N, b = e - a * N;
    const int wa = L.ao[a], wb = L.ao[b];
    const int sa = ao_sigma(L, wa), sb = ao_sigma(L, wb);
    K[e] = D[(size_t)sa * N + sb] + D2[(size_t)sb * N + sa];
    const int whi = a >= b ? wa : wb, wlo = a >= b ? wb : wa;                // the pair (hi >= lo) in original order
    const int shi = a >= b ? sa : sb;
    const int c = ao_cls(wa) ^ ao_cls(wb);
    const long long q = bl_cbase(L, c) + bl_fullsec(L, c, ao_cls(whi)) + L.kinfo[(size_t)c * N + shi].offA + ao_loc(wlo);
    const int h2 = max(sa, sb), l2 = min(sa, sb);
    const int r = rowmap[(size_t)h2 * (h2 + 1) / 2 + l2];
    double s = 0.0;
    if (r >= 0)
        for (int w = 0; w < NW; ++w)
            for (int pl = 0; pl < MP; ++pl)
                if (L.kap0[c * NW + w] + pl * KS < L.cntA[(L.chunk_cls[w] ^ c) * N + shi]) s += Jd[pl * planeJd + (size_t)r * NW + w];
    for (int t = 0; t < nseg; ++t) s += Jt[(size_t)t * L.NPtot + q];
    J[e] = s;
}

// ---- writing the tensor -------------------------------------------------------------------------------------------------------
// The generation slab holds, per Cartesian bra component pair (ca, cb) of class c, a row of the COMPLETE-row shape of class c
// (pair index without cbase; stride RLS = max_c NP[c]): the ket-transformed values (ca cb|kl) for the pairs (k >= l) of class c.
struct OutRowP {
    int i, j;              // output AO indices (i >= j), original order: rows of the Cartesian -> spherical CSR
    int iI, lamj;          // internal index of i; loc of j
    int c, ncb;            // class of the row; components of the second bra shell
    int cartA, cartB;      // first Cartesian AO of the two bra shells
    int secoff[4];         // start of section a in the stored row
    int len;               // doubles of the row
    int unr, upos, pad;    // rows of the row's storage unit, its position in it
    long long slab_off;    // first slab row of this bra pair
    long long ubase;       // base of the unit in the stored tensor
};

// tensor row (i,j) = bra transform of the slab rows; pad slots and the slots beyond l == j in the segment of k == i <- 0.
// grid (ceil(max row length / 256), rows); the (<= 6 x 6) Cartesian -> spherical terms of the two bra AOs are staged in LDS once
// per block (nested sums, fixed order).
__global__ __launch_bounds__(256) void xform_bra_store_packed(const double *__restrict__ in, double *__restrict__ eri,
                                                              const OutRowP *__restrict__ rows, long long RLS, BLayout L,
                                                              const int *__restrict__ ptr, const int *__restrict__ idx,
                                                              const double *__restrict__ val)
{
    __shared__ double sValA[32], sValB[32];
    __shared__ long long sOffA[32], sOffB[32];
    const OutRowP R = rows[blockIdx.y];
    if ((int)(blockIdx.x * 256) >= R.len) return;
    const int pa = ptr[R.i], na = min(32, ptr[R.i + 1] - pa), pb = ptr[R.j], nb = min(32, ptr[R.j + 1] - pb);
    if (threadIdx.x < na) { sValA[threadIdx.x] = val[pa + threadIdx.x]; sOffA[threadIdx.x] = (long long)(idx[pa + threadIdx.x] - R.cartA) * R.ncb * RLS; }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + nb) {
        const int t = threadIdx.x - 64;
        sValB[t] = val[pb + t]; sOffB[t] = (long long)(idx[pb + t] - R.cartB) * RLS;
    }
    __syncthreads();
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= R.len) return;
    // section of x: the sections follow each other in internal class order (ascending cstart)
    int a = 0, best = -1, besta = -1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int so = R.secoff[t], st = bl_cstart(L, t);
        if (so <= x && (so > best || (so == best && st > besta))) { best = so; besta = st; a = t; }
    }
    const int c = R.c;
    const int srcidx = x - best + bl_fullsec(L, c, a);
    const int kI = L.gk[bl_gbase(L, c) + srcidx / TF_SEG_PAD];
    const KInfo ki = L.kinfo[(size_t)c * L.N + kI];
    const int lam = srcidx - bl_fullsec(L, c, a) - ki.offA;
    double s = 0.0;
    if (lam < ki.cnt && !(kI == R.iI && lam > R.lamj)) {
        const double *__restrict__ src = in + R.slab_off * RLS + srcidx;
        for (int qa = 0; qa < na; ++qa) {
            double t = 0.0;
            for (int qb = 0; qb < nb; ++qb) t += sValB[qb] * src[sOffA[qa] + sOffB[qb]];
            s += sValA[qa] * t;
        }
    }
    const int pc = (ki.cnt + TF_SEG_PAD - 1) & ~(TF_SEG_PAD - 1);
    eri[R.ubase + (long long)R.unr * (best + ki.offA) + (long long)R.upos * pc + lam] = s;
}

// (ij|kl) from the stored tensor, original indices; rows owned by another rank read as 0 (sum over ranks = the tensor).
// rowoff[r]: base of the storage unit of local row r; rowsec[6 r ..]: secoff[4], position in the unit, rows of the unit
__device__ __forceinline__ double packed_element(const double *__restrict__ eri, const int *__restrict__ rowmap,
                                                 const long long *__restrict__ rowoff, const int *__restrict__ rowsec, const BLayout &L,
                                                 int i, int j, int k, int l)
{
    int ih = max(i, j), il = min(i, j), kh = max(k, l), kl = min(k, l);
    const int wi = L.ao[ih], wj = L.ao[il], wk = L.ao[kh], wl = L.ao[kl];
    const int c = ao_cls(wi) ^ ao_cls(wj);
    if (c != (ao_cls(wk) ^ ao_cls(wl))) return 0.0;                          // x/y parity, pyx:1324-1327
    const long long p = (long long)ih * (ih + 1) / 2 + il, q = (long long)kh * (kh + 1) / 2 + kl;
    int wr1 = wi, wr2 = wj, wc1 = wk, wc2 = wl;                              // row pair (the larger one), column pair
    if (q > p) { wr1 = wk; wr2 = wl; wc1 = wi; wc2 = wj; }
    const int s1 = ao_sigma(L, wr1), s2 = ao_sigma(L, wr2);
    const int h2 = max(s1, s2), l2 = min(s1, s2);
    const int r = rowmap[(size_t)h2 * (h2 + 1) / 2 + l2];
    if (r < 0) return 0.0;
    const int a = ao_cls(wc1);
    const KInfo ki = L.kinfo[(size_t)c * L.N + ao_sigma(L, wc1)];
    const int pc = (ki.cnt + TF_SEG_PAD - 1) & ~(TF_SEG_PAD - 1);
    const int *rs = rowsec + 6 * (size_t)r;                                  // secoff[4], position in the unit, rows of the unit
    return eri[rowoff[r] + (long long)rs[5] * (rs[a] + ki.offA) + (long long)rs[4] * pc + ao_loc(wc2)];
}

// packed -> dense N^4 with all images (what the reference leaves in ERI_AO, pyx:1335-1342).
__global__ void expand_dense_packed_kernel(const double *__restrict__ eri, const int *__restrict__ rowmap, const long long *__restrict__ rowoff,
                                           const int *__restrict__ rowsec, BLayout L, double *__restrict__ dense)
{
    const int N = L.N;
    const long long total = (long long)N * N * N * N;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(e % N);
        long long r = e / N;
        const int k = (int)(r % N); r /= N;
        const int j = (int)(r % N);
        const int i = (int)(r / N);
        dense[e] = packed_element(eri, rowmap, rowoff, rowsec, L, i, j, k, l);
    }
}

__global__ void sample_packed_kernel(const double *__restrict__ eri, const int *__restrict__ rowmap, const long long *__restrict__ rowoff,
                                     const int *__restrict__ rowsec, BLayout L, long long n, const int *__restrict__ idx, double *__restrict__ out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    out[q] = packed_element(eri, rowmap, rowoff, rowsec, L, idx[4 * q], idx[4 * q + 1], idx[4 * q + 2], idx[4 * q + 3]);
}

// The stored part of local rows as symmetric matrices, for the GEMM-shaped consumers (AO->MO).  With L the tensor restricted to the
// pairs (kl) <= (ij) (its diagonal (kl) == (ij) halved), (ij|kl) = L + L^T in the pair indices: every stored value is read once, from
// its own row, and a rank needs nothing but its own rows.  out[r - r0][k][l] = out[r - r0][l][k] = L[(ij), (kl)] for the pairs the row
// holds; the caller has zeroed out.  ORIGINAL indices, leading dimension ld.  Grid: (ceil(max NP / 256), rows of the slab).
__global__ void unpack_own_rows_kernel(const double *__restrict__ eri, const long long *__restrict__ rowoff, const int *__restrict__ rowsec,
                                       BLayout L, const int2 *__restrict__ row_ij, long long r0, int ld, double *__restrict__ out)
{
    const long long r = r0 + blockIdx.y;
    const int2 ij = row_ij[r];
    const int wi = L.ao[ij.x], wj = L.ao[ij.y];
    const int c = ao_cls(wi) ^ ao_cls(wj), iI = ao_sigma(L, wi), lamj = ao_loc(wj);
    const int x = blockIdx.x * 256 + threadIdx.x;                            // pair index inside a complete class-c row
    if (x >= bl_np(L, c)) return;
    const int kI = L.gk[bl_gbase(L, c) + x / TF_SEG_PAD];
    const int a = L.clsI[kI];
    const KInfo ki = L.kinfo[(size_t)c * L.N + kI];
    const int lam = x - bl_fullsec(L, c, a) - ki.offA;
    if (lam >= ki.cnt || kI - bl_cstart(L, a) >= L.cntA[(size_t)a * L.N + iI] || (kI == iI && lam > lamj)) return;   // padding; k > i; (kl) > (ij)
    const int *rs = rowsec + 6 * (size_t)r;
    const int pc = (ki.cnt + TF_SEG_PAD - 1) & ~(TF_SEG_PAD - 1);
    double v = eri[rowoff[r] + (long long)rs[5] * (rs[a] + ki.offA) + (long long)rs[4] * pc + lam];
    if (kI == iI && lam == lamj) v *= 0.5;
    const int k = L.origI[kI], l = L.origI[bl_cstart(L, a ^ c) + lam];
    double *__restrict__ o = out + (size_t)blockIdx.y * L.N * ld;
    o[(size_t)k * ld + l] = v;
    o[(size_t)l * ld + k] = v;
}

// The same in class-blocked form, for the rows of ONE class c (the list `rows`): a row of class c is nonzero only in the four blocks
// (k of class a) x (l of class a ^ c), so only those are materialised -- a quarter of the N x N matrix.  Block a of a row: |a| x ldb[a]
// doubles at boff[a] (ldb[a] >= |a ^ c|), INTERNAL class-local indices (loc); row stride rstride.  The caller has zeroed out.
struct RowBlocks { int boff[4], ldb[4]; long long rstride; };
__global__ void unpack_own_rows_blocked_kernel(const double *__restrict__ eri, const long long *__restrict__ rowoff, const int *__restrict__ rowsec,
                                               BLayout L, const int2 *__restrict__ row_ij, const int *__restrict__ rows, int c, RowBlocks RBk,
                                               double *__restrict__ out)
{
    const long long r = rows[blockIdx.y];
    const int2 ij = row_ij[r];
    const int wi = L.ao[ij.x], wj = L.ao[ij.y];
    const int iI = ao_sigma(L, wi), lamj = ao_loc(wj);
    const int x = blockIdx.x * 256 + threadIdx.x;                            // pair index inside a complete class-c row
    if (x >= bl_np(L, c)) return;
    const int kI = L.gk[bl_gbase(L, c) + x / TF_SEG_PAD];
    const int a = L.clsI[kI], b = a ^ c;
    const KInfo ki = L.kinfo[(size_t)c * L.N + kI];
    const int lam = x - bl_fullsec(L, c, a) - ki.offA;
    const int kl = kI - bl_cstart(L, a);                                     // loc of k
    if (lam >= ki.cnt || kl >= L.cntA[(size_t)a * L.N + iI] || (kI == iI && lam > lamj)) return;   // padding; k > i; (kl) > (ij)
    const int *rs = rowsec + 6 * (size_t)r;
    const int pc = (ki.cnt + TF_SEG_PAD - 1) & ~(TF_SEG_PAD - 1);
    double v = eri[rowoff[r] + (long long)rs[5] * (rs[a] + ki.offA) + (long long)rs[4] * pc + lam];
    if (kI == iI && lam == lamj) v *= 0.5;
    double *__restrict__ o = out + (size_t)blockIdx.y * RBk.rstride;
    o[RBk.boff[a] + (size_t)kl * RBk.ldb[a] + lam] = v;                      // [k][l] in block a
    o[RBk.boff[b] + (size_t)lam * RBk.ldb[b] + kl] = v;                      // [l][k] in block b (the same slot when k == l)
}

// out[x][y] = G1[x][y] + G2[y][x]   (x < A, y < B; G1 [A][B], G2 [B][A]): the two halves L and L^T of a transformed tensor
__global__ void add_transposed_kernel(const double *__restrict__ G1, const double *__restrict__ G2, long long A, long long B, double *__restrict__ out)
{
    __shared__ double tile[32][33];
    const long long x0 = (long long)blockIdx.y * 32, y0 = (long long)blockIdx.x * 32;
    for (int t = threadIdx.y; t < 32; t += blockDim.y) {                     // tile of G2: rows y0.., columns x0..
        const long long y = y0 + t, x = x0 + threadIdx.x;
        tile[t][threadIdx.x] = (y < B && x < A) ? G2[y * A + x] : 0.0;
    }
    __syncthreads();
    for (int t = threadIdx.y; t < 32; t += blockDim.y) {
        const long long x = x0 + t, y = y0 + threadIdx.x;
        if (x < A && y < B) out[x * B + y] = G1[x * B + y] + tile[threadIdx.x][t];
    }
}
