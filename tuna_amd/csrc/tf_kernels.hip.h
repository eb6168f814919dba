// tf_kernels.hip.h -- gfx950 kernels of libtunafock: ERI generation, Cartesian->spherical slab
// transforms, the fused J/K pass and small helpers.  Included once by tf_device.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "tf_internal.h"

namespace tfk {

struct DShell { int L, ncomp, comp_off, cart_off; };
struct DPair { int A, B, La, Lb, npp, pp_off, nE, pad; long long e_off; };

struct DBasis {
    const DShell *shells;
    const DPair *pairs;
    const int8_t *c_lx, *c_ly, *c_lz;
    const double *c_scale;
    const double *pp_p, *pp_Pz, *pp_K;
    const double *epool;
    const double *boys;      // [NGRID][NORD]
};

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

// One Cartesian component quartet of a shell quartet, decoded once per thread.
struct CompQuartet {
    int lx12, ly12, lz12, lx34, ly34, lz34;
    int ixab, iyab, izab, ixcd, iycd, izcd;
    int ca, cb, cc, cd;
    bool nonzero;
    double cscale;
};

// ------------------------------------------------------------------------------------------------
// K1: contracted Cartesian integrals of one shell quartet (AB|CD) per workgroup.
//   grid.x = ket shell pair (all pairs C >= D), grid.y = bra shell pair of the current slab.
//   Primitive quartets are processed in batches of PB (as many Boys/R tables as fit the LDS budget):
//   phase 1 builds the tables -- cooperatively (L+1 lanes per table, one barrier per row) when the batch is
//   small, one thread per table when there are many; phase 2 -- threads own Cartesian component quartets and
//   contract the Hermite expansion tables (global, L1/L2 resident) with the R tables (LDS).
//   When all primitive quartets fit one batch (every uncontracted shell quartet) the tables are built once
//   and reused by every component chunk.
//   Output: rows (ca,cb) of the Cartesian slab C[row][Nc][Nc], positions [k][l] and [l][k].
// Reference: primitive_pair_eri pyx:1142-1221, contraction pyx:1235-1253, driver pyx:1314-1342.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TF_ERI_THREADS) void eri_cart_kernel(DBasis B, const int *__restrict__ bra_pairs,
                                                                  const long long *__restrict__ bra_rowoff, int Nc,
                                                                  double *__restrict__ Cslab)
{
    __shared__ double sR[TF_RT_DOUBLES];
    __shared__ double sPref[TF_ERI_THREADS];
    __shared__ double sPQ[TF_ERI_THREADS];
    __shared__ double sRed[TF_ERI_THREADS];

    const int tid = threadIdx.x;
    const DPair ab = B.pairs[bra_pairs[blockIdx.y]];
    const DPair cd = B.pairs[blockIdx.x];
    const DShell sa = B.shells[ab.A], sb = B.shells[ab.B], sc = B.shells[cd.A], sd = B.shells[cd.B];
    const int L = ab.La + ab.Lb + cd.La + cd.Lb;
    const int tsize = (L + 1) * (L + 2) / 2;
    int PB = TF_RT_DOUBLES / tsize - 1;
    if (PB > TF_ERI_THREADS) PB = TF_ERI_THREADS;
    const int stride = PB | 1;
    const int npq = ab.npp * cd.npp;
    const int ncomp = sa.ncomp * sb.ncomp * sc.ncomp * sd.ncomp;
    const int nEab = ab.nE, nEcd = cd.nE;
    const int Lab1 = ab.La + ab.Lb + 1, Lcd1 = cd.La + cd.Lb + 1;
    const double *__restrict__ Eab0 = B.epool + ab.e_off;
    const double *__restrict__ Ecd0 = B.epool + cd.e_off;
    const long long row0 = bra_rowoff[blockIdx.y];
    const size_t NcNc = (size_t)Nc * Nc;
    const bool one_batch = npq <= PB;

    // ---- phase 1: Boys + R tables of primitive quartets [b0, b0+nb) into LDS columns 0..nb-1 (all threads call) ----
    auto phase1 = [&](int b0, int nb) {
        const int L1 = L + 1;
        if (nb * L1 <= TF_ERI_THREADS) {
            const int q = tid / L1, n = tid - q * L1;
            const bool mine = q < nb;
            if (mine && n == 0) {
                const int pq = b0 + q;
                const int pab = pq / cd.npp, pcd = pq - pab * cd.npp;
                const double p = B.pp_p[ab.pp_off + pab], qq = B.pp_p[cd.pp_off + pcd];
                const double s = p + qq, alpha = p * qq / s;
                const double PQ = B.pp_Pz[ab.pp_off + pab] - B.pp_Pz[cd.pp_off + pcd];
                build_R_row0(sR, stride, q, L, alpha, PQ, B.boys);
                sPQ[q] = PQ;
                sPref[q] = B.pp_K[ab.pp_off + pab] * B.pp_K[cd.pp_off + pcd] * (34.986836655249725 / (p * qq * sqrt(s)));
            }
            for (int v = 1; v <= L; ++v) {
                __syncthreads();
                if (mine && n <= L - v) {
                    const int r0 = tri_index(v, 0, L), r1 = tri_index(v - 1, 0, L);
                    double val = sPQ[q] * sR[(r1 + n + 1) * stride + q];
                    if (v > 1) val += (double)(v - 1) * sR[(tri_index(v - 2, 0, L) + n + 1) * stride + q];
                    sR[(r0 + n) * stride + q] = val;
                }
            }
        } else if (tid < nb) {
            const int pq = b0 + tid;
            const int pab = pq / cd.npp, pcd = pq - pab * cd.npp;
            const double p = B.pp_p[ab.pp_off + pab], qq = B.pp_p[cd.pp_off + pcd];
            const double s = p + qq, alpha = p * qq / s;
            const double PQ = B.pp_Pz[ab.pp_off + pab] - B.pp_Pz[cd.pp_off + pcd];
            build_R_column(sR, stride, tid, L, alpha, PQ, B.boys);
            // 2 pi^(5/2) / (p q sqrt(p+q)) * coefficient product, pyx:1219-1221
            sPref[tid] = B.pp_K[ab.pp_off + pab] * B.pp_K[cd.pp_off + pcd] * (34.986836655249725 / (p * qq * sqrt(s)));
        }
    };

    // ---- decode the component quartet `c` of this shell quartet ----
    auto decode = [&](int c, CompQuartet &Q) {
        Q.cd = c % sd.ncomp; c /= sd.ncomp;
        Q.cc = c % sc.ncomp; c /= sc.ncomp;
        Q.cb = c % sb.ncomp; Q.ca = c / sb.ncomp;
        const int ia = sa.comp_off + Q.ca, ib = sb.comp_off + Q.cb, ic = sc.comp_off + Q.cc, id = sd.comp_off + Q.cd;
        const int ax = B.c_lx[ia], ay = B.c_ly[ia], az = B.c_lz[ia];
        const int bx = B.c_lx[ib], by = B.c_ly[ib], bz = B.c_lz[ib];
        const int cx = B.c_lx[ic], cy = B.c_ly[ic], cz = B.c_lz[ic];
        const int dx = B.c_lx[id], dy = B.c_ly[id], dz = B.c_lz[id];
        Q.lx12 = ax + bx; Q.ly12 = ay + by; Q.lz12 = az + bz;
        Q.lx34 = cx + dx; Q.ly34 = cy + dy; Q.lz34 = cz + dz;
        Q.ixab = (ax * (ab.Lb + 1) + bx) * Lab1; Q.iyab = (ay * (ab.Lb + 1) + by) * Lab1; Q.izab = (az * (ab.Lb + 1) + bz) * Lab1;
        Q.ixcd = (cx * (cd.Lb + 1) + dx) * Lcd1; Q.iycd = (cy * (cd.Lb + 1) + dy) * Lcd1; Q.izcd = (cz * (cd.Lb + 1) + dz) * Lcd1;
        Q.nonzero = !(((Q.lx12 + Q.lx34) & 1) || ((Q.ly12 + Q.ly34) & 1));      // x/y parity, pyx:1324-1327
        Q.cscale = B.c_scale[ia] * B.c_scale[ib] * B.c_scale[ic] * B.c_scale[id];
        if ((Q.lx34 + Q.ly34) & 1) Q.cscale = -Q.cscale;                          // (-1)^(tau+nu) is fixed by parity
    };

    // ---- phase 2: my component against primitive quartets g, g+NG, ... of the batch in LDS ----
    auto phase2 = [&](const CompQuartet &Q, int b0, int nb, int g, int NG) -> double {
        double acc = 0.0;
        for (int qq = g; qq < nb; qq += NG) {
            const int pq = b0 + qq;
            const int pab = pq / cd.npp, pcd = pq - pab * cd.npp;
            const double *__restrict__ Exy12 = Eab0 + (size_t)pab * 2 * nEab;
            const double *__restrict__ Ez12 = Exy12 + nEab;
            const double *__restrict__ Exy34 = Ecd0 + (size_t)pcd * 2 * nEcd;
            const double *__restrict__ Ez34 = Exy34 + nEcd;
            const double *__restrict__ Rq = sR + qq;
            double sum = 0.0;
            for (int t = Q.lx12 & 1; t <= Q.lx12; t += 2) {
                const double ex12 = Exy12[Q.ixab + t];
                for (int tau = Q.lx34 & 1; tau <= Q.lx34; tau += 2) {
                    const double xf = ex12 * Exy34[Q.ixcd + tau] * c_dfact[(t + tau) >> 1];
                    for (int u = Q.ly12 & 1; u <= Q.ly12; u += 2) {
                        const double ey12 = Exy12[Q.iyab + u];
                        for (int nu = Q.ly34 & 1; nu <= Q.ly34; nu += 2) {
                            const double xyf = xf * ey12 * Exy34[Q.iycd + nu] * c_dfact[(u + nu) >> 1];
                            const int nxy = ((t + tau) >> 1) + ((u + nu) >> 1);
                            double zs = 0.0;
                            for (int v = 0; v <= Q.lz12; ++v) {
                                const double ez12 = Ez12[Q.izab + v];
                                double zphi = 0.0;
                                for (int phi = 0; phi <= Q.lz34; ++phi) {
                                    const double r = Rq[(tri_index(v + phi, nxy, L)) * stride];
                                    const double e34 = Ez34[Q.izcd + phi];
                                    zphi += (phi & 1) ? -(e34 * r) : (e34 * r);
                                }
                                zs += ez12 * zphi;
                            }
                            sum += xyf * zs;
                        }
                    }
                }
            }
            acc += sPref[qq] * sum;
        }
        return acc;
    };

    if (one_batch) {
        phase1(0, npq);
        __syncthreads();
    }
    for (int chunk0 = 0; chunk0 < ncomp; chunk0 += TF_ERI_THREADS) {
        const int nchunk = min(TF_ERI_THREADS, ncomp - chunk0);
        // thread -> (group g, component c0): groups split the primitive quartets of a batch
        int ncp = 1;
        while (ncp < nchunk) ncp <<= 1;
        const int NG = TF_ERI_THREADS / ncp;
        const int g = tid / ncp, c0 = tid - g * ncp;
        const bool active = c0 < nchunk;
        CompQuartet Q;
        Q.nonzero = false; Q.cscale = 0.0; Q.ca = Q.cb = Q.cc = Q.cd = 0;
        if (active) decode(chunk0 + c0, Q);
        double acc = 0.0;
        if (one_batch) {
            if (active && Q.nonzero) acc = phase2(Q, 0, npq, g, NG);
        } else {
            for (int b0 = 0; b0 < npq; b0 += PB) {
                const int nb = min(PB, npq - b0);
                __syncthreads();
                phase1(b0, nb);
                __syncthreads();
                if (active && Q.nonzero) acc += phase2(Q, b0, nb, g, NG);
            }
        }
        // combine the groups (fixed order -> bitwise reproducible)
        if (NG > 1) {
            __syncthreads();
            sRed[tid] = acc;
            __syncthreads();
            if (g == 0 && active) {
                double s = 0.0;
                for (int gg = 0; gg < NG; ++gg) s += sRed[gg * ncp + c0];
                acc = s;
            }
        }
        if (g == 0 && active) {
            const double val = acc * Q.cscale;
            const size_t row = (size_t)(row0 + (long long)Q.ca * sb.ncomp + Q.cb);
            const int k = sc.cart_off + Q.cc, l = sd.cart_off + Q.cd;
            Cslab[row * NcNc + (size_t)k * Nc + l] = val;
            if (cd.A != cd.B) Cslab[row * NcNc + (size_t)l * Nc + k] = val;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Slab transforms (reference: transform_to_spherical_harmonics, kernel:504-523, done there as two
// sparse kron(U,U) products over the whole tensor).  U is block diagonal and very sparse, so each
// output element is a short CSR dot product.
// ------------------------------------------------------------------------------------------------

// out[r][k][ls] = sum_e val_e * in[r][k][idx_e]      (last axis: Nc -> Ns)
__global__ void xform_last_axis(const double *__restrict__ in, double *__restrict__ out, long long nrow_k, int Nc,
                                int Ns, const int *__restrict__ ptr, const int *__restrict__ idx,
                                const double *__restrict__ val)
{
    const long long total = nrow_k * Ns;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long rk = e / Ns;
        const int ls = (int)(e - rk * Ns);
        const double *__restrict__ src = in + rk * Nc;
        double s = 0.0;
        for (int q = ptr[ls]; q < ptr[ls + 1]; ++q) s += val[q] * src[idx[q]];
        out[e] = s;
    }
}

// out[r][ks][l] (leading dim ld, zero padded) = sum_e val_e * in[r][idx_e][l]     (middle axis: Nc -> Ns)
__global__ void xform_mid_axis(const double *__restrict__ in, double *__restrict__ out, long long nrow, int Nc, int Ns,
                               int ld, const int *__restrict__ ptr, const int *__restrict__ idx,
                               const double *__restrict__ val)
{
    const long long per_row = (long long)Ns * ld, total = nrow * per_row;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / per_row;
        const int rem = (int)(e - r * per_row);
        const int ks = rem / ld, l = rem - ks * ld;
        double s = 0.0;
        if (l < Ns) {
            const double *__restrict__ src = in + r * (long long)Nc * Ns + l;
            for (int q = ptr[ks]; q < ptr[ks + 1]; ++q) s += val[q] * src[(long long)idx[q] * Ns];
        }
        out[e] = s;
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

template <int NLC, int JB>
__global__ __launch_bounds__(TF_JK_THREADS) void jk_rows_kernel(const double *__restrict__ eri,
                                                                const int2 *__restrict__ row_ij, long long n_rows, int N,
                                                                int ld, const double *__restrict__ P /*[N][ld]*/,
                                                                double *__restrict__ Jrow, double *__restrict__ Kp)
{
    extern __shared__ double smem[];
    double *sPj = smem;                 // [JB][N]  P[:, j_b]
    double *sPi = smem + JB * N;        // [JB][N]  P[:, i_b]
    double *sRed = smem + 2 * JB * N;   // reduction scratch: 4 * TF_JK_THREADS doubles

    const long long row0 = (long long)blockIdx.x * JB;
    const int tid = threadIdx.x;
    int2 ij[JB];
    bool valid[JB];
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        valid[b] = row0 + b < n_rows;
        ij[b] = valid[b] ? row_ij[row0 + b] : make_int2(0, 0);
    }
    for (int k = tid; k < N; k += TF_JK_THREADS) {
#pragma unroll
        for (int b = 0; b < JB; ++b) {
            sPj[b * N + k] = P[(size_t)k * ld + ij[b].y];
            sPi[b * N + k] = P[(size_t)k * ld + ij[b].x];
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

    double accJ[JB];
    double2 k1[JB][NLC], k2[JB][NLC];
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        accJ[b] = 0.0;
#pragma unroll
        for (int c = 0; c < NLC; ++c) { k1[b][c] = make_double2(0.0, 0.0); k2[b][c] = make_double2(0.0, 0.0); }
    }

    if (active) {
#pragma unroll 2
        for (int k = tk; k < N; k += TK) {
            const double2 *__restrict__ Pk = reinterpret_cast<const double2 *>(P + (size_t)k * ld);
#pragma unroll
            for (int c = 0; c < NLC; ++c) {
                const int lp = tl + c * TF_JK_THREADS;
                if (NLC == 1 || lp < npair) {
                    const double2 p = Pk[lp];
                    double2 m[JB];
#pragma unroll
                    for (int b = 0; b < JB; ++b)
                        m[b] = valid[b] ? load_stream(reinterpret_cast<const double2 *>(M0 + b * row_len + (size_t)k * ld) + lp)
                                        : make_double2(0.0, 0.0);
#pragma unroll
                    for (int b = 0; b < JB; ++b) {
                        const double pj = sPj[b * N + k], pi = sPi[b * N + k];
                        accJ[b] += m[b].x * p.x + m[b].y * p.y;
                        k1[b][c].x += m[b].x * pj; k1[b][c].y += m[b].y * pj;
                        k2[b][c].x += m[b].x * pi; k2[b][c].y += m[b].y * pi;
                    }
                }
            }
        }
    }
    // J: block reduction per row (fixed tree)
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        __syncthreads();
        sRed[tid] = accJ[b];
        __syncthreads();
        for (int s = TF_JK_THREADS / 2; s > 0; s >>= 1) {
            if (tid < s) sRed[tid] += sRed[tid + s];
            __syncthreads();
        }
        if (tid == 0 && valid[b]) Jrow[row0 + b] = sRed[0];
    }
    // K partials: sum over tk for each column pair
    double2 *sK = reinterpret_cast<double2 *>(sRed);
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        double *Kp1 = Kp + (size_t)(row0 + b) * 2 * ld, *Kp2 = Kp1 + ld;
#pragma unroll
        for (int c = 0; c < NLC; ++c) {
            const int lp = tl + c * TF_JK_THREADS;
            if (TK == 1) {                                     // one thread per column pair: nothing to combine
                if (valid[b] && lp < npair) {
                    reinterpret_cast<double2 *>(Kp1)[lp] = k1[b][c];
                    reinterpret_cast<double2 *>(Kp2)[lp] = k2[b][c];
                }
                continue;
            }
            __syncthreads();
            if (active) { sK[tid] = k1[b][c]; sK[TF_JK_THREADS + tid] = k2[b][c]; }
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
