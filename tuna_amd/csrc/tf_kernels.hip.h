// tf_kernels.hip.h -- gfx950 kernels of libtunafock: ERI generation, Cartesian->spherical slab
// transforms, the fused J/K pass and small helpers.  Included once by tf_device.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "tf_internal.h"
#include "tf_layout.hip.h"
#include "tf_dbasis.hip.h"

namespace tfk {

#define TF_ERI_THREADS 256
#define TF_RT_DOUBLES 6400   // LDS doubles for the per-batch R tables

// (2k-1)!! for k = 0..10  (the closed form of the x/y Hermite-Coulomb integrals at zero x/y separation,
// reference: odd_double_fact_even_argument_fast, pyx:914-947)
__device__ __constant__ double c_dfact[11] = {1.0, 1.0, 3.0, 15.0, 105.0, 945.0, 10395.0, 135135.0,
                                               2027025.0, 34459425.0, 654729075.0};

__device__ __forceinline__ int tri_index(int v, int n, int L) { return v * (L + 1) - (v * (v - 1)) / 2 + n; }

// Boys F_0..F_L(T) and the z-only Hermite-Coulomb table R[v][n] (n <= L - v) for ONE primitive quartet,
// written by one thread into its LDS column `col` (element idx lives at R[idx * stride + col]).
// Reference: fill_boys_table pyx:1540-1572, fill_pow_table pyx:1582-1602, fill_Rz_linear_table pyx:1612-1651.
__device__ __forceinline__ void build_R_column(double *R, int stride, int col, int L, double alpha, double PQ,
                                               const double *__restrict__ boys)
{
    const double T = alpha * PQ * PQ;
    double *F = R + col;   // row v = 0 occupies idx 0..L
    if (T == 0.0) {
        for (int m = 0; m <= L; ++m) F[m * stride] = 1.0 / (2.0 * m + 1.0);
    } else if (T < TF_BOYS_TMAX) {
        const int i = (int)(T * (1.0 / TF_BOYS_STEP) + 0.5);
        const double d = (double)i * TF_BOYS_STEP - T;     // F_m(T) = sum_k F_{m+k}(T0) d^k / k!
        const double *row = boys + (size_t)i * TF_BOYS_NORD + L;
        double f = row[8];
        f = row[7] + f * d * (1.0 / 8.0);
        f = row[6] + f * d * (1.0 / 7.0);
        f = row[5] + f * d * (1.0 / 6.0);
        f = row[4] + f * d * (1.0 / 5.0);
        f = row[3] + f * d * (1.0 / 4.0);
        f = row[2] + f * d * (1.0 / 3.0);
        f = row[1] + f * d * (1.0 / 2.0);
        f = row[0] + f * d;
        const double e = exp(-T), two_T = 2.0 * T;
        F[L * stride] = f;
        for (int m = L; m > 0; --m) {                        // downward recursion, pyx:1570-1572
            f = (two_T * f + e) / (2.0 * m - 1.0);
            F[(m - 1) * stride] = f;
        }
    } else {
        // T >= 36: erf(sqrt T) = 1 to double precision; upward recursion is contracting for m < T
        const double e = exp(-T), inv2T = 1.0 / (2.0 * T);
        double f = 0.5 * sqrt(3.141592653589793238462643383279 / T);
        F[0] = f;
        for (int m = 0; m < L; ++m) {
            f = ((2.0 * m + 1.0) * f - e) * inv2T;
            F[(m + 1) * stride] = f;
        }
    }
    // R[0][n] = (-2 alpha)^n F_n
    double pw = 1.0;
    const double fac = -2.0 * alpha;
    for (int n = 0; n <= L; ++n) { F[n * stride] *= pw; pw *= fac; }
    // R[v][n] = PQ R[v-1][n+1] + (v-1) R[v-2][n+1]
    for (int v = 1; v <= L; ++v) {
        const int r0 = tri_index(v, 0, L), r1 = tri_index(v - 1, 0, L), r2 = (v > 1) ? tri_index(v - 2, 0, L) : 0;
        for (int n = L - v; n >= 0; --n) {
            double val = PQ * R[(r1 + n + 1) * stride + col];
            if (v > 1) val += (double)(v - 1) * R[(r2 + n + 1) * stride + col];
            R[(r0 + n) * stride + col] = val;
        }
    }
}

// Boys values F_0..F_L(T), already multiplied by (-2 alpha)^n, into row 0 of column `col` (leader part of the
// cooperative table build below; same arithmetic as build_R_column).
__device__ __forceinline__ void build_R_row0(double *R, int stride, int col, int L, double alpha, double PQ,
                                             const double *__restrict__ boys)
{
    const double T = alpha * PQ * PQ;
    double *F = R + col;
    if (T == 0.0) {
        for (int m = 0; m <= L; ++m) F[m * stride] = 1.0 / (2.0 * m + 1.0);
    } else if (T < TF_BOYS_TMAX) {
        const int i = (int)(T * (1.0 / TF_BOYS_STEP) + 0.5);
        const double d = (double)i * TF_BOYS_STEP - T;
        const double *row = boys + (size_t)i * TF_BOYS_NORD + L;
        double f = row[8];
        f = row[7] + f * d * (1.0 / 8.0);
        f = row[6] + f * d * (1.0 / 7.0);
        f = row[5] + f * d * (1.0 / 6.0);
        f = row[4] + f * d * (1.0 / 5.0);
        f = row[3] + f * d * (1.0 / 4.0);
        f = row[2] + f * d * (1.0 / 3.0);
        f = row[1] + f * d * (1.0 / 2.0);
        f = row[0] + f * d;
        const double e = exp(-T), two_T = 2.0 * T;
        F[L * stride] = f;
        for (int m = L; m > 0; --m) {
            f = (two_T * f + e) / (2.0 * m - 1.0);
            F[(m - 1) * stride] = f;
        }
    } else {
        const double e = exp(-T), inv2T = 1.0 / (2.0 * T);
        double f = 0.5 * sqrt(3.141592653589793238462643383279 / T);
        F[0] = f;
        for (int m = 0; m < L; ++m) {
            f = ((2.0 * m + 1.0) * f - e) * inv2T;
            F[(m + 1) * stride] = f;
        }
    }
    double pw = 1.0;
    const double fac = -2.0 * alpha;
    for (int n = 0; n <= L; ++n) { F[n * stride] *= pw; pw *= fac; }
}

// ------------------------------------------------------------------------------------------------
// Slab transforms (reference: transform_to_spherical_harmonics, kernel:504-523, done there as two
// sparse kron(U,U) products over the whole tensor).  U is block diagonal and very sparse, so each
// output element is a short CSR dot product.
// ------------------------------------------------------------------------------------------------

// Both ket axes in one pass: out[r][ks][ls] (leading dim ld) = sum_a U[ks][a] (sum_b U[ls][b] in[r][a][b]) -- no intermediate array
// (a version with one kernel per axis made two extra passes over the slab: 3.4 -> 2.0 ms for Ar2/cc-pVQZ, 26 -> 17 ms for N2/cc-pV5Z).
// grid (ceil(Ns / 4), rows): one output row ks per wave, lanes over ls.  tri: only ls <= ks is needed (packed layout).
__global__ __launch_bounds__(256) void xform_ket_both(const double *__restrict__ in, double *__restrict__ out, int Nc, int Ns, int ld,
                                                      const int *__restrict__ ptr, const int *__restrict__ idx,
                                                      const double *__restrict__ val, int tri)
{
    __shared__ double sVal[4][32];
    __shared__ int sIdx[4][32];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ks = 4 * blockIdx.x + w;
    const long long r = blockIdx.y;
    int na = 0;
    if (ks < Ns) {
        const int p0 = ptr[ks];
        na = min(32, ptr[ks + 1] - p0);
        if (lane < na) { sVal[w][lane] = val[p0 + lane]; sIdx[w][lane] = idx[p0 + lane]; }
    }
    __syncthreads();
    if (ks >= Ns) return;
    const double *__restrict__ src = in + r * (long long)Nc * Nc;
    double *__restrict__ dst = out + (r * Ns + ks) * (long long)ld;
    const int lend = tri ? ks + 1 : ld;
    for (int l = lane; l < lend; l += 64) {
        double s = 0.0;
        if (l < Ns) {
            const int q0 = ptr[l], q1 = ptr[l + 1];
            for (int qa = 0; qa < na; ++qa) {
                const double *__restrict__ rowp = src + (long long)sIdx[w][qa] * Nc;
                double t = 0.0;
                for (int q = q0; q < q1; ++q) t += val[q] * rowp[idx[q]];
                s += sVal[w][qa] * t;
            }
        }
        dst[l] = s;
    }
}

// The same for the packed layout: slab row r (a Cartesian bra component pair of parity class rowcls[r]) keeps only the pairs
// (ks >= ls) of that class, in the complete-row shape (tf_jkpacked.hip.h): out[r][fullsec + offA(ks) + loc(ls)].  A quarter of the
// outputs of xform_ket_both, no zeros stored.  grid (ceil(Ns / 4), rows): one output AO ks per wave, lanes over the segment.
__global__ __launch_bounds__(256) void xform_ket_packed(const double *__restrict__ in, double *__restrict__ out, int Nc, BLayout L, long long RLS,
                                                        const signed char *__restrict__ rowcls, const int *__restrict__ ptr,
                                                        const int *__restrict__ idx, const double *__restrict__ val)
{
    __shared__ double sVal[4][32];
    __shared__ int sIdx[4][32];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ks = 4 * blockIdx.x + w, Ns = L.N;
    const long long r = blockIdx.y;
    int na = 0;
    if (ks < Ns) {
        const int p0 = ptr[ks];
        na = min(32, ptr[ks + 1] - p0);
        if (lane < na) { sVal[w][lane] = val[p0 + lane]; sIdx[w][lane] = idx[p0 + lane]; }
    }
    __syncthreads();
    if (ks >= Ns) return;
    const int c = rowcls[r], wk = L.ao[ks], ck = ao_cls(wk), b = ck ^ c;
    const KInfo ki = L.kinfo[(size_t)c * Ns + ao_sigma(L, wk)];
    const double *__restrict__ src = in + r * (long long)Nc * Nc;
    double *__restrict__ dst = out + r * RLS + bl_fullsec(L, c, ck) + ki.offA;
    for (int lam = lane; lam < ki.cnt; lam += 64) {
        const int l = L.origI[bl_cstart(L, b) + lam];
        const int q0 = ptr[l], q1 = ptr[l + 1];
        double s = 0.0;
        for (int qa = 0; qa < na; ++qa) {
            const double *__restrict__ rowp = src + (long long)sIdx[w][qa] * Nc;
            double t = 0.0;
            for (int q = q0; q < q1; ++q) t += val[q] * rowp[idx[q]];
            s += sVal[w][qa] * t;
        }
        dst[lam] = s;
    }
}

struct OutRow {
    int i, j;              // output AO indices (i >= j)
    int cartA, cartB;      // first Cartesian AO of the two bra shells
    int ncb, pad;
    long long slab_off;    // first slab row of this bra pair
    long long dst_row;     // row in the stored tensor
};

// tensor row (i,j) = sum_{ea in U row i} sum_{eb in U row j} va vb * slab[(ca,cb)]      (bra axes)
__global__ void xform_bra_store(const double *__restrict__ in, double *__restrict__ eri, const OutRow *__restrict__ rows,
                                long long row_len, const int *__restrict__ ptr, const int *__restrict__ idx,
                                const double *__restrict__ val)
{
    const OutRow R = rows[blockIdx.y];
    const double *__restrict__ src = in + R.slab_off * row_len;
    double *__restrict__ dst = eri + R.dst_row * row_len;
    for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < row_len; x += (long long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int qa = ptr[R.i]; qa < ptr[R.i + 1]; ++qa) {
            const long long ra = (long long)(idx[qa] - R.cartA) * R.ncb;
            double t = 0.0;
            for (int qb = ptr[R.j]; qb < ptr[R.j + 1]; ++qb)
                t += val[qb] * src[(ra + (idx[qb] - R.cartB)) * row_len + x];
            s += val[qa] * t;
        }
        dst[x] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Fock build from the stored tensor (reference: scf:70 "ijkl,kl->ij" and scf:42 "ilkj,kl->ij").
// Stored rows are (i >= j) x full [k][l] (leading dimension ld).  One pass over a row M = (ij|..) gives
//     J_ij = J_ji = <M, P>,   K_i. += M^T P[:,j],   K_j. += M^T P[:,i]   (second one only when i != j)
// so every stored byte is read exactly once per build.  A workgroup streams JB consecutive rows together so
// that each P[k][l] it pulls from L2 serves JB rows; M is read with non-temporal 16-byte loads (touched once,
// must not evict P from L2).  Per-row partial K vectors go to a scratch buffer and are summed in a fixed order
// by jk_reduce_kernel (no atomics: results are bitwise reproducible).
// ------------------------------------------------------------------------------------------------
#define TF_JK_THREADS 256

__device__ __forceinline__ double2 load_stream(const double2 *p)
{
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
}

// ND densities per pass (1 = RHF, 2 = UHF alpha/beta): the same bytes of M serve ND times the flops.
template <int NLC, int JB, int ND>
__global__ __launch_bounds__(TF_JK_THREADS) void jk_rows_kernel(const double *__restrict__ eri,
                                                                const int2 *__restrict__ row_ij, long long n_rows, int N,
                                                                int ld, const double *__restrict__ P0 /*[N][ld]*/,
                                                                const double *__restrict__ P1, double *__restrict__ Jrow,
                                                                double *__restrict__ Kp)
{
    extern __shared__ double smem[];
    double *sPj = smem;                      // [ND][JB][N]  P_d[:, j_b]
    double *sPi = smem + ND * JB * N;        // [ND][JB][N]  P_d[:, i_b]
    double *sRed = smem + 2 * ND * JB * N;   // reduction scratch: 4 * TF_JK_THREADS doubles

    const long long row0 = (long long)blockIdx.x * JB;
    const int tid = threadIdx.x;
    const double *__restrict__ Pd[2] = {P0, ND > 1 ? P1 : P0};
    int2 ij[JB];
    bool valid[JB];
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        valid[b] = row0 + b < n_rows;
        ij[b] = valid[b] ? row_ij[row0 + b] : make_int2(0, 0);
    }
    for (int k = tid; k < N; k += TF_JK_THREADS) {
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int b = 0; b < JB; ++b) {
                sPj[(d * JB + b) * N + k] = Pd[d][(size_t)k * ld + ij[b].y];
                sPi[(d * JB + b) * N + k] = Pd[d][(size_t)k * ld + ij[b].x];
            }
    }
    __syncthreads();

    const int npair = ld >> 1;                                  // double2 columns per line
    const int TL = (npair < TF_JK_THREADS) ? npair : TF_JK_THREADS;
    const int TK = TF_JK_THREADS / TL;
    const int tk = tid / TL, tl = tid - tk * TL;
    const bool active = tk < TK;
    const size_t row_len = (size_t)N * ld;
    const double *__restrict__ M0 = eri + (size_t)row0 * row_len;

    double accJ[ND][JB];
    double2 k1[ND][JB][NLC], k2[ND][JB][NLC];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int b = 0; b < JB; ++b) {
            accJ[d][b] = 0.0;
#pragma unroll
            for (int c = 0; c < NLC; ++c) { k1[d][b][c] = make_double2(0.0, 0.0); k2[d][b][c] = make_double2(0.0, 0.0); }
        }

    if (active) {
#pragma unroll 2
        for (int k = tk; k < N; k += TK) {
#pragma unroll
            for (int c = 0; c < NLC; ++c) {
                const int lp = tl + c * TF_JK_THREADS;
                if (NLC == 1 || lp < npair) {
                    double2 p[ND];
#pragma unroll
                    for (int d = 0; d < ND; ++d) p[d] = reinterpret_cast<const double2 *>(Pd[d] + (size_t)k * ld)[lp];
                    double2 m[JB];
#pragma unroll
                    for (int b = 0; b < JB; ++b)
                        m[b] = valid[b] ? load_stream(reinterpret_cast<const double2 *>(M0 + b * row_len + (size_t)k * ld) + lp)
                                        : make_double2(0.0, 0.0);
#pragma unroll
                    for (int d = 0; d < ND; ++d)
#pragma unroll
                        for (int b = 0; b < JB; ++b) {
                            const double pj = sPj[(d * JB + b) * N + k], pi = sPi[(d * JB + b) * N + k];
                            accJ[d][b] += m[b].x * p[d].x + m[b].y * p[d].y;
                            k1[d][b][c].x += m[b].x * pj; k1[d][b][c].y += m[b].y * pj;
                            k2[d][b][c].x += m[b].x * pi; k2[d][b][c].y += m[b].y * pi;
                        }
                }
            }
        }
    }
    double2 *sK = reinterpret_cast<double2 *>(sRed);
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        double *Jrow_d = Jrow + (size_t)d * n_rows;
        double *Kp_d = Kp + (size_t)d * n_rows * 2 * ld;
        // J: block reduction per row (fixed tree)
#pragma unroll
        for (int b = 0; b < JB; ++b) {
            __syncthreads();
            sRed[tid] = accJ[d][b];
            __syncthreads();
            for (int s = TF_JK_THREADS / 2; s > 0; s >>= 1) {
                if (tid < s) sRed[tid] += sRed[tid + s];
                __syncthreads();
            }
            if (tid == 0 && valid[b]) Jrow_d[row0 + b] = sRed[0];
        }
        // K partials: sum over tk for each column pair
#pragma unroll
        for (int b = 0; b < JB; ++b) {
            double *Kp1 = Kp_d + (size_t)(row0 + b) * 2 * ld, *Kp2 = Kp1 + ld;
#pragma unroll
            for (int c = 0; c < NLC; ++c) {
                const int lp = tl + c * TF_JK_THREADS;
                if (TK == 1) {                                     // one thread per column pair: nothing to combine
                    if (valid[b] && lp < npair) {
                        reinterpret_cast<double2 *>(Kp1)[lp] = k1[d][b][c];
                        reinterpret_cast<double2 *>(Kp2)[lp] = k2[d][b][c];
                    }
                    continue;
                }
                __syncthreads();
                if (active) { sK[tid] = k1[d][b][c]; sK[TF_JK_THREADS + tid] = k2[d][b][c]; }
                __syncthreads();
                if (tk == 0 && lp < npair && valid[b]) {
                    double2 u = make_double2(0.0, 0.0), w = make_double2(0.0, 0.0);
                    for (int q = 0; q < TK; ++q) {
                        const double2 x = sK[q * TL + tl], y = sK[TF_JK_THREADS + q * TL + tl];
                        u.x += x.x; u.y += x.y; w.x += y.x; w.y += y.y;
                    }
                    reinterpret_cast<double2 *>(Kp1)[lp] = u;
                    reinterpret_cast<double2 *>(Kp2)[lp] = w;
                }
            }
        }
    }
}

// J[i][l], K[i][l] from the per-row partials; workgroup = (i, 64-column chunk), 4 groups of 64 lanes split the j sum
// (fixed combination order).  rowmap[i(i+1)/2+j] = local row or -1 (row owned by another rank).
__global__ __launch_bounds__(256) void jk_reduce_kernel(const double *__restrict__ Jrow, const double *__restrict__ Kp,
                                                        const int *__restrict__ rowmap, int N, int ld, double *__restrict__ J,
                                                        double *__restrict__ K)
{
    __shared__ double sPart[256];
    const int i = blockIdx.x;
    const int lane = threadIdx.x & 63, jg = threadIdx.x >> 6;
    const int l = blockIdx.y * 64 + lane;
    double s = 0.0;
    if (l < N) {
#pragma unroll 4
        for (int j = jg; j < N; j += 4) {
            const int hi = max(i, j), lo = min(i, j);
            const int r = rowmap[hi * (hi + 1) / 2 + lo];
            if (r >= 0) s += Kp[(size_t)r * 2 * ld + (j <= i ? 0 : ld) + l];
        }
    }
    sPart[threadIdx.x] = s;
    __syncthreads();
    if (jg == 0 && l < N) {
        K[(size_t)i * N + l] = ((sPart[lane] + sPart[64 + lane]) + sPart[128 + lane]) + sPart[192 + lane];
        const int hi = max(i, l), lo = min(i, l);
        const int r = rowmap[hi * (hi + 1) / 2 + lo];
        J[(size_t)i * N + l] = (r >= 0) ? Jrow[r] : 0.0;
    }
}

// dense P [N][N] -> padded [N][ld]
__global__ void pad_matrix_kernel(const double *__restrict__ in, double *__restrict__ out, int N, int ld)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * ld) return;
    const int r = e / ld, c = e - r * ld;
    out[e] = (c < N) ? in[(size_t)r * N + c] : 0.0;
}

// stored rows -> dense N^4 with all images (what the reference leaves in ERI_AO, pyx:1335-1342)
__global__ void expand_dense_kernel(const double *__restrict__ eri, const int *__restrict__ rowmap, int N, int ld,
                                    double *__restrict__ dense)
{
    const long long total = (long long)N * N * N * N;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const int l = (int)(e % N);
        long long r = e / N;
        const int k = (int)(r % N); r /= N;
        const int j = (int)(r % N);
        const int i = (int)(r / N);
        const int hi = max(i, j), lo = min(i, j);
        const int row = rowmap[hi * (hi + 1) / 2 + lo];
        dense[e] = (row >= 0) ? eri[((size_t)row * N + k) * ld + l] : 0.0;
    }
}

__global__ void sample_kernel(const double *__restrict__ eri, const int *__restrict__ rowmap, int N, int ld,
                              long long n, const int *__restrict__ idx, double *__restrict__ out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const int i = idx[4 * q], j = idx[4 * q + 1], k = idx[4 * q + 2], l = idx[4 * q + 3];
    const int hi = max(i, j), lo = min(i, j);
    const int row = rowmap[hi * (hi + 1) / 2 + lo];
    out[q] = (row >= 0) ? eri[((size_t)row * N + k) * ld + l] : 0.0;
}

}  // namespace tfk
