// tf_eri_team.hip.h -- ERI generation, "team" kernels (round 3): one shell quartet per TEAM of lanes, several teams per workgroup.
// Reference: primitive_pair_eri pyx:1142-1221 (the 6-deep Hermite sum), contraction pyx:1235-1253, driver and parity rule
// pyx:1267-1355; ket half of transform_to_spherical_harmonics kernel:504-523.
//
// A launch holds the shell quartets of ONE class (La, Lb | Lc, Ld): grid (ket groups, bra pairs), a workgroup = one bra shell pair
// x NT = 256 / TEAM consecutive ket pairs of the class list.  TEAM = 16 or 64 lanes of one wavefront (the small and middle classes: the
// team's tables are private to its wave, so nothing but wave-level ordering of LDS operations is needed after the shared staging
// barrier) or the whole workgroup (the top classes, whose tables need it).  The template parameters are the two pair sums
// LAB = La + Lb and LCD = Lc + Ld: every loop over Hermite indices t, tau, v, phi and over the Boys order has compile-time bounds and
// the table rows of a lane live in registers.
//
// The reference's sum separates on a z-axis diatomic (x and y separations are zero) into per-axis tables of the exponent 4-tuples:
//     X[ax,bx,cx,dx][m] = (2m-1)!! sum_{t + tau = 2m} Ex12[t] Ex34[tau] (-1)^tau           (x and y share the tables)
//     Z[az,bz,cz,dz][n] = sum_v Ez12[v] G[cz,dz][v][n],   G[c,d][v][n] = sum_phi (-1)^phi Ez34[phi] R[v + phi][n]
//     (ab|cd)           = pref sum_{m + m' < NM} X[x tuple][m] X[y tuple][m'] Z[z tuple][m + m']
// Phases of a team: (0) stage the ket Hermite tables, signs folded in; (1) Boys values F_n(T), one order per lane (Taylor expansion on
// the tabulated grid), and the z-only Hermite-Coulomb table R[v][n] row by row; (2) G, one (ket tuple, v) row per lane; (3) X and Z, one
// (bra tuple, ket tuple) row per lane; (4) the Cartesian components that are not zero by x/y parity (pyx:1324-1327), class by class,
// into an LDS block; (5) the ket half of the Cartesian -> spherical transform as a sparse pair transform inside each parity class,
// written straight to the half-transformed slab row of the Cartesian bra component pair (complete-row shape of the packed layout).
#pragma once
#include <hip/hip_runtime.h>
#include "tf_dbasis.hip.h"

namespace tfk {

struct TClass {
    int La, Lb, Lc, Ld;
    int nTab, nTcd, nT;           // exponent tuples of one axis: (La+1)(Lb+1), (Lc+1)(Ld+1), their product
    float inv_nTcd;
    int nab, ncd;                 // Cartesian component pairs of the bra / ket shell pair
    int pA[5], pK[5], pS[5];      // parity-class offsets: bra component pairs, ket component pairs, ket output pairs (class-sorted lists)
    int nkap, nnzT;               // ket output pairs; entries of the ket pair transform
    int tabA, tabK;               // DBasis::ct_* offsets of a representative bra / ket pair (the tables are the same for a whole class)
    int ktp_off, kte_off;         // ket pair transform of the ket class: row pointers at kt_ptr[ktp_off ..], entries at kt_k / kt_c[kte_off ..]
    int n_ket;                    // ket pairs of the launch (class list prefix)
    int nEab, nEcd;               // doubles of one Hermite table of the bra / ket pair
    int vcap;                     // doubles of a team's component block (eri_teamc_kernel: of its table scratch)
    int nacc;                     // parity-allowed Cartesian components of a quartet (eri_teamc_kernel: the accumulation block)
    int oE12, oOffA, oScA, oOffK, oTp, oTk, oTc, shared_doubles, team_doubles;   // LDS carve-out, in doubles
    // flat component / output lists (classes whose parity-allowed components fit the team's block: one loop over all of them instead
    // of one per parity class): DBasis::tflat[flat_off ..] = nacc words (bra index | ket index << 16, class-sorted indices), then nout
    // pairs (bra index | output pair << 16, offset of the bra row inside the block)
    int flat, flat_off, nflat, nout, oCompW, oOutW, oRowOff;   // nflat: words to stage (components padded to an even count, then outputs)
    float invK[4], invS[4];       // 1 / (ket component pairs of class c), 1 / (ket output pairs of class c)
    long long RLS;                // stride of a slab row
};

// What a team needs of a ket / bra shell pair (uncontracted: one primitive pair), parallel to the launch's pair lists: no pointer chasing
// through DBasis::pairs.  Kq = weight product / exponent sum (the 1 / (p q) of the prefactor folded in on the host).
struct KetRec { double q, Qz, Kq; unsigned e_off; int kq; };                                  // kq: first entry of the pair's slab offsets in kq_off
struct BraRec { double p, Pz, Kp; unsigned e_off; int A; long long row_first; long long pad; };   // row_first: slab row of the pair's first component pair

template <int TEAM>
__device__ __forceinline__ void wave_lds_order()
{
    // LDS operations of one wave execute in order; this only stops the compiler from moving them across
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int TEAM>
__device__ __forceinline__ void team_sync()
{
    if constexpr (TEAM > 64) __syncthreads();
    else wave_lds_order<TEAM>();
}

// floor(x / d) for small non-negative x with inv = 1 / d in single precision ((x + 1/2) / d is never within 1/(2d) of an integer)
__device__ __forceinline__ int small_div(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

template <int NM>
__device__ __forceinline__ double team_fact_sum(const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z)
{
    double x[NM], y[NM], z[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) { x[m] = X[m]; y[m] = Y[m]; z[m] = Z[m]; }
    double sum = 0.0;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        double t = 0.0;
#pragma unroll
        for (int mp = 0; mp + m < NM; ++mp) t += y[mp] * z[m + mp];
        sum += x[m] * t;
    }
    return sum;
}

// Phases 1-3 of a team for ONE primitive quartet: Boys values and the z-only Hermite-Coulomb table R, the ket half G of the z tables,
// then the X and Z rows.  On entry sE34 holds the ket Hermite tables (signs folded in) and sR is zeroed; sE12 the bra tables.  Ends
// with a team barrier: sX / sZ are complete, sE34 / sR / sG dead.
template <int LAB, int LCD, int TEAM>
__device__ __forceinline__ void team_tables(const DBasis &B, int tl, double alpha, double PQ, double T, const double *sE12, int nEab,
                                            const double *sE34, int nEcd, double *sR, double *sG, double *sX, double *sZ, int nTcd, int nT,
                                            float inv_nTcd)
{
    constexpr int L = LAB + LCD, NM = L / 2 + 1, XS = NM | 1, Lab1 = LAB + 1, Lcd1 = LCD + 1, RS = L + 2;
    constexpr double DF[11] = {1.0, 1.0, 3.0, 15.0, 105.0, 945.0, 10395.0, 135135.0, 2027025.0, 34459425.0, 654729075.0};
    team_sync<TEAM>();
    // ---- phase 1: Boys values (reference: fill_boys_table pyx:1540-1572) and R[v][n] = PQ R[v-1][n+1] + (v-1) R[v-2][n+1] (pyx:1612-1651) ----
    if (tl <= L) {
        const int n = tl;
        double f;
        if (T < TF_BOYS_TMAX) {
            const int i = (int)(T * (1.0 / TF_BOYS_STEP) + 0.5);
            const double d = (double)i * TF_BOYS_STEP - T;            // F_n(T) = sum_k F_{n+k}(T0) d^k / k!
            const double *__restrict__ row = B.boys + (size_t)i * TF_BOYS_NORD + n;
            f = row[8];
            f = row[7] + f * d * (1.0 / 8.0);
            f = row[6] + f * d * (1.0 / 7.0);
            f = row[5] + f * d * (1.0 / 6.0);
            f = row[4] + f * d * (1.0 / 5.0);
            f = row[3] + f * d * (1.0 / 4.0);
            f = row[2] + f * d * (1.0 / 3.0);
            f = row[1] + f * d * (1.0 / 2.0);
            f = row[0] + f * d;
        } else {
            // T >= 36: erf(sqrt T) = 1 to double precision; the upward recursion is contracting for m < T
            const double e = exp(-T), inv2T = 1.0 / (2.0 * T);
            double g = 0.5 * sqrt(3.141592653589793238462643383279 / T);
            f = g;
#pragma unroll
            for (int m = 0; m < L; ++m) {
                g = ((2.0 * m + 1.0) * g - e) * inv2T;
                if (m + 1 == n) f = g;
            }
        }
        double pw = 1.0, fac = -2.0 * alpha;                               // (-2 alpha)^n by binary powering (n <= 12)
#pragma unroll
        for (int bit = 0; (1 << bit) <= L; ++bit) {
            pw = ((n >> bit) & 1) ? pw * fac : pw;
            fac *= fac;
        }
        sR[n] = f * pw;                                                    // R[0][n] = (-2 alpha)^n F_n
    }
#pragma unroll
    for (int v = 1; v <= L; ++v) {
        wave_lds_order<TEAM>();
        if (tl <= L - v) {
            double val = PQ * sR[(v - 1) * RS + tl + 1];
            if (v > 1) val += (double)(v - 1) * sR[(v - 2) * RS + tl + 1];
            sR[v * RS + tl] = val;
        }
    }
    team_sync<TEAM>();
    // ---- phase 2: G[cdt][v][n] = sum_phi Ez34'[phi] R[v + phi][n]   (entries beyond n + v + phi <= L meet a zero coefficient or a zero of R) ----
    for (int e = tl; e < nTcd * Lab1; e += TEAM) {
        const int cdt = e / Lab1, v = e - cdt * Lab1;
        double ez[Lcd1];
#pragma unroll
        for (int ph = 0; ph < Lcd1; ++ph) ez[ph] = sE34[nEcd + cdt * Lcd1 + ph];
        const double *Rv = sR + v * RS;
#pragma unroll
        for (int n = 0; n < NM; ++n) {
            double g = 0.0;
#pragma unroll
            for (int ph = 0; ph < Lcd1; ++ph) g += ez[ph] * Rv[ph * RS + n];
            sG[e * NM + n] = g;
        }
    }
    team_sync<TEAM>();
    // ---- phase 3: X and Z rows, one exponent tuple pair per lane ----
    for (int tu = tl; tu < nT; tu += TEAM) {
        const int abt = small_div(tu, inv_nTcd), cdt = tu - abt * nTcd;
        double e12x[Lab1], e12z[Lab1], e34x[Lcd1];
#pragma unroll
        for (int t = 0; t < Lab1; ++t) { e12x[t] = sE12[abt * Lab1 + t]; e12z[t] = sE12[nEab + abt * Lab1 + t]; }
#pragma unroll
        for (int t = 0; t < Lcd1; ++t) e34x[t] = sE34[cdt * Lcd1 + t];
        const double *Gc = sG + cdt * Lab1 * NM;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            double x = 0.0;
#pragma unroll
            for (int t = 0; t < Lab1; ++t)
                if (2 * m - t >= 0 && 2 * m - t <= LCD) x += e12x[t] * e34x[2 * m - t];
            sX[tu * XS + m] = x * DF[m];
            double z = 0.0;
#pragma unroll
            for (int v = 0; v < Lab1; ++v) z += e12z[v] * Gc[v * NM + m];
            sZ[tu * XS + m] = z;
        }
    }
    team_sync<TEAM>();
}

#ifndef TF_TEAM_OCC
#define TF_TEAM_OCC 1            // minimum waves per SIMD the team kernels are compiled for (register budget)
#endif
template <int LAB, int LCD, int TEAM>
__global__ __launch_bounds__(256, TF_TEAM_OCC) void eri_team_kernel(DBasis B, TClass tc, const BraRec *__restrict__ bras, const KetRec *__restrict__ kets,
                                                       const int *__restrict__ kcnt, double *__restrict__ T2)
{
    constexpr int NT = 256 / TEAM, L = LAB + LCD, NM = L / 2 + 1, XS = NM | 1, Lab1 = LAB + 1, Lcd1 = LCD + 1, RS = L + 2;
    constexpr double DF[11] = {1.0, 1.0, 3.0, 15.0, 105.0, 945.0, 10395.0, 135135.0, 2027025.0, 34459425.0, 654729075.0};
    static_assert(L + 1 <= TEAM && L + 1 <= 64, "the R table is built by the first lanes of the team's first wave");
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int team = TEAM >= 256 ? 0 : tid / TEAM, tl = TEAM >= 256 ? tid : tid % TEAM;
    const BraRec *__restrict__ ab = bras + blockIdx.y;
    // kets of the class list this bra pair needs: those whose first shell does not lie above the bra's (the list ascends in it)
    const int nket_b = min(tc.n_ket, kcnt[ab->A]);
    if ((int)blockIdx.x * NT >= nket_b) return;

    // ---- shared staging: bra Hermite tables, component-pair tables of both shell pairs, the ket pair transform ----
    double *sE12 = smem + tc.oE12, *sScA = smem + tc.oScA, *sTc = smem + tc.oTc;
    int4 *sOffA = reinterpret_cast<int4 *>(smem + tc.oOffA), *sOffK = reinterpret_cast<int4 *>(smem + tc.oOffK);
    int *sTp = reinterpret_cast<int *>(smem + tc.oTp), *sTk = reinterpret_cast<int *>(smem + tc.oTk);
    const int nEab = tc.nEab, nEcd = tc.nEcd, nTcd = tc.nTcd;
    {
        const double *__restrict__ gEab = B.epool + ab->e_off;
        for (int k = tid; k < 2 * nEab; k += 256) sE12[k] = gEab[k];
        const int unitA = nTcd * XS * (int)sizeof(double), unitK = XS * (int)sizeof(double);
        for (int s = tid; s < tc.nab; s += 256) {
            const int f = B.ct_ord[tc.tabA + s], w = B.ct_ix[tc.tabA + f];
            sOffA[s] = make_int4((w & 255) * unitA, ((w >> 8) & 255) * unitA, ((w >> 16) & 255) * unitA, f);
            sScA[s] = B.ct_sc[tc.tabA + f];
        }
        for (int s = tid; s < tc.ncd; s += 256) {
            const int f = B.ct_ord[tc.tabK + s], w = B.ct_ix[tc.tabK + f];
            sOffK[s] = make_int4((w & 255) * unitK, ((w >> 8) & 255) * unitK, ((w >> 16) & 255) * unitK, f);
        }
        for (int s = tid; s <= tc.nkap; s += 256) sTp[s] = B.kt_ptr[tc.ktp_off + s];
        for (int s = tid; s < tc.nnzT; s += 256) { sTk[s] = B.kt_k[tc.kte_off + s]; sTc[s] = B.kt_c[tc.kte_off + s]; }
        if (tc.flat) {
            int *sW = reinterpret_cast<int *>(smem + tc.oCompW);          // component words, then output words (contiguous)
            for (int s = tid; s < tc.nflat; s += 256) sW[s] = B.tflat[tc.flat_off + s];
        }
    }
    const long long row_first = ab->row_first;
    long long *sRowOff = reinterpret_cast<long long *>(smem + tc.oRowOff);    // slab offset of the row of bra component pair s (class-sorted index)
    for (int s = tid; s < tc.nab; s += 256) sRowOff[s] = (row_first + B.ct_ord[tc.tabA + s]) * tc.RLS;
    __syncthreads();

    const double p = ab->p, Pz = ab->Pz, Kp = ab->Kp;
    double *tm = smem + tc.shared_doubles + team * tc.team_doubles;
    double *sX = tm, *sZ = sX + tc.nT * XS, *scr = sZ + tc.nT * XS;
    double *sE34 = scr, *sR = sE34 + 2 * nEcd, *sG = sR + (L + 1) * RS;
    double *sV = scr;                                                     // the component block lives over E34 / R / G once X and Z exist
    int *sDoff = reinterpret_cast<int *>(scr + tc.vcap);
    const char *bX = reinterpret_cast<const char *>(sX), *bZ = reinterpret_cast<const char *>(sZ);

    // ---- the workgroup's ket groups: a team takes ket g NT + team of every group g = blockIdx.x, blockIdx.x + gridDim.x, ... ----
    for (int g = blockIdx.x; g * NT < nket_b; g += gridDim.x) {
    const int kq = g * NT + team;
    if (TEAM >= 256 || kq < nket_b) {
    const KetRec *__restrict__ cd = kets + kq;
    const double q = cd->q;
    const double s = p + q, rs = rsqrt(s), pq = p * q, alpha = pq * (rs * rs);
    const double PQ = Pz - cd->Qz;
    const double T = alpha * PQ * PQ;
    // 2 pi^(5/2) / (p q sqrt(p+q)) * coefficient product, pyx:1219-1221 (1 / p and 1 / q are folded into Kp, Kq)
    const double pref = Kp * cd->Kq * (34.986836655249725 * rs);

    // ---- phase 0: ket Hermite tables with (-1)^tau, (-1)^phi folded in; R zeroed; slab offsets of the output pairs ----
    {
        const double *__restrict__ gEcd = B.epool + cd->e_off;
        for (int k = tl; k < 2 * nEcd; k += TEAM) {
            const double v = gEcd[k];
            sE34[k] = ((k % Lcd1) & 1) ? -v : v;                           // (both tables have rows of Lcd1 entries; nEcd is a multiple of Lcd1)
        }
        for (int k = tl; k < (L + 1) * RS; k += TEAM) sR[k] = 0.0;
        for (int k = tl; k < tc.nkap; k += TEAM) sDoff[k] = B.kq_off[cd->kq + k];   // (behind the scratch area: untouched by the phases)
    }
    team_tables<LAB, LCD, TEAM>(B, tl, alpha, PQ, T, sE12, nEab, sE34, nEcd, sR, sG, sX, sZ, nTcd, tc.nT, tc.inv_nTcd);

    if (tc.flat) {
        // ---- phases 4 and 5 over the flat lists: every parity-allowed component, then every output, in one loop each ----
        const int *sCompW = reinterpret_cast<const int *>(smem + tc.oCompW);
        const int2 *sOutW = reinterpret_cast<const int2 *>(smem + tc.oOutW);
        for (int j = tl; j < tc.nacc; j += TEAM) {
            const unsigned w = (unsigned)sCompW[j];
            const int4 oa = sOffA[w & 0xffffu], oc = sOffK[w >> 16];
            const double *X = reinterpret_cast<const double *>(bX + (oa.x + oc.x));
            const double *Y = reinterpret_cast<const double *>(bX + (oa.y + oc.y));
            const double *Z = reinterpret_cast<const double *>(bZ + (oa.z + oc.z));
            sV[j] = team_fact_sum<NM>(X, Y, Z);
        }
        team_sync<TEAM>();
        for (int o = tl; o < tc.nout; o += TEAM) {
            const int2 w = sOutW[o];
            const int iA = w.x & 0xffff, kap = (int)((unsigned)w.x >> 16);
            const int doff = sDoff[kap];
            if (doff < 0) continue;
            const double *Vr = sV + w.y;
            double acc = 0.0;
            for (int e = sTp[kap]; e < sTp[kap + 1]; ++e) acc += sTc[e] * Vr[sTk[e]];
            T2[sRowOff[iA] + doff] = acc * (pref * sScA[iA]);
        }
        team_sync<TEAM>();
    } else
    // ---- phases 4 and 5, parity class by parity class, in chunks of complete bra rows ----
    for (int c = 0; c < 4; ++c) {
        const int nA = tc.pA[c + 1] - tc.pA[c], nK = tc.pK[c + 1] - tc.pK[c], nS = tc.pS[c + 1] - tc.pS[c];
        if (nA == 0 || nK == 0 || nS == 0) continue;
        const float invK = tc.invK[c], invS = tc.invS[c];
        const int rows = max(1, tc.vcap / nK);
        const int4 *offA = sOffA + tc.pA[c], *offK = sOffK + tc.pK[c];
        const double *scA = sScA + tc.pA[c];
        for (int i0 = 0; i0 < nA; i0 += rows) {
            const int ni = min(rows, nA - i0);
            for (int j = tl; j < ni * nK; j += TEAM) {
                const int il = small_div(j, invK), kl = j - il * nK;
                const int4 oa = offA[i0 + il], oc = offK[kl];
                const double *X = reinterpret_cast<const double *>(bX + (oa.x + oc.x));
                const double *Y = reinterpret_cast<const double *>(bX + (oa.y + oc.y));
                const double *Z = reinterpret_cast<const double *>(bZ + (oa.z + oc.z));
                sV[j] = team_fact_sum<NM>(X, Y, Z);
            }
            team_sync<TEAM>();
            for (int o = tl; o < ni * nS; o += TEAM) {
                const int il = small_div(o, invS), kap = tc.pS[c] + (o - il * nS);
                const int doff = sDoff[kap];
                if (doff < 0) continue;
                const double *Vr = sV + il * nK;
                double acc = 0.0;
                for (int e = sTp[kap]; e < sTp[kap + 1]; ++e) acc += sTc[e] * Vr[sTk[e]];
                T2[sRowOff[tc.pA[c] + i0 + il] + doff] = acc * (pref * scA[i0 + il]);
            }
            team_sync<TEAM>();
        }
    }
    }   // this team's ket
    }   // ket groups
}

}  // namespace tfk
