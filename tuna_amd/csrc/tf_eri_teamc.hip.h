// tf_eri_teamc.hip.h -- team ERI kernel for CONTRACTED shell quartets and for task lists that mix classes (the small-problem mode: a
// real basis set has too few quartets per class for per-class launches to fill the chip).
// Reference: contraction loop calculate_electron_repulsion_integral_cached pyx:1235-1253 around primitive_pair_eri pyx:1142-1221.
//
// Same phases as eri_team_kernel (tf_eri_team.hip.h), with two differences: (1) a workgroup takes a TASK -- (class record, bra shell
// pair, up to NT ket pairs of one class) -- from a list, so ONE launch per template instance (LAB, LCD, TEAM) serves every class with
// those pair sums; (2) a team loops over the primitive quartets (pab, pcd) of its shell quartet: tables per primitive quartet (phases
// 0-3), the parity-allowed Cartesian components accumulated in an LDS block (phase 4), and after the loop the ket pair transform and
// the stores (phase 5).  Quartets with more than `pq_max` primitive quartets are left to eri_cfact_kernel (same predicate there).
#pragma once
#include <hip/hip_runtime.h>
#include "tf_eri_team.hip.h"

namespace tfk {

struct TeamTask {
    int cls;                      // index of the class record
    int bra;                      // bra shell pair
    int ket0, nk;                 // kets: klist[ket0 .. ket0 + nk), nk <= 256 / TEAM
    long long row_first;          // slab row of the bra pair's first component pair
};

template <int LAB, int LCD, int TEAM>
__global__ __launch_bounds__(256, TF_TEAM_OCC) void eri_teamc_kernel(DBasis B, const TClass *__restrict__ tcs, const TeamTask *__restrict__ tasks,
                                                                     const int *__restrict__ klist, int pq_max, double *__restrict__ T2)
{
    constexpr int L = LAB + LCD, NM = L / 2 + 1, XS = NM | 1, Lcd1 = LCD + 1, RS = L + 2;
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int team = TEAM >= 256 ? 0 : tid / TEAM, tl = TEAM >= 256 ? tid : tid % TEAM;
    const TeamTask task = tasks[blockIdx.x];
    const TClass &tc = tcs[task.cls];
    const DPair *__restrict__ ab = B.pairs + task.bra;

    // ---- shared staging: component-pair tables of both shell pairs, the ket pair transform ----
    double *sScA = smem + tc.oScA, *sTc = smem + tc.oTc;
    int4 *sOffA = reinterpret_cast<int4 *>(smem + tc.oOffA), *sOffK = reinterpret_cast<int4 *>(smem + tc.oOffK);
    int *sTp = reinterpret_cast<int *>(smem + tc.oTp), *sTk = reinterpret_cast<int *>(smem + tc.oTk);
    const int nEab = tc.nEab, nEcd = tc.nEcd, nTcd = tc.nTcd;
    {
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
    }
    __syncthreads();
    if (team >= task.nk) return;
    const int pcd_pair = klist[task.ket0 + team];
    const DPair *__restrict__ cd = B.pairs + pcd_pair;
    const int npp_ab = ab->npp, npp_cd = cd->npp;
    if (npp_ab * npp_cd > pq_max) return;                                 // (eri_cfact_kernel computes this one)

    double *tm = smem + tc.shared_doubles + team * tc.team_doubles;
    double *sX = tm, *sZ = sX + tc.nT * XS, *scr = sZ + tc.nT * XS;
    double *sE12 = scr, *sE34 = sE12 + 2 * nEab, *sR = sE34 + 2 * nEcd, *sG = sR + (L + 1) * RS;
    double *sAcc = scr + tc.vcap;                                         // all parity-allowed components of the quartet, class by class
    int *sDoff = reinterpret_cast<int *>(sAcc + tc.nacc);
    const char *bX = reinterpret_cast<const char *>(sX), *bZ = reinterpret_cast<const char *>(sZ);

    for (int k = tl; k < tc.nacc; k += TEAM) sAcc[k] = 0.0;
    for (int k = tl; k < tc.nkap; k += TEAM) sDoff[k] = B.kq_off[B.kq_ptr[pcd_pair] + k];
    const double *__restrict__ gEab = B.epool + ab->e_off;
    const double *__restrict__ gEcd = B.epool + cd->e_off;
    for (int pab = 0; pab < npp_ab; ++pab) {
        const double p = B.pp_p[ab->pp_off + pab], Pz = B.pp_Pz[ab->pp_off + pab], Kab = B.pp_K[ab->pp_off + pab];
        // (sE12 / sE34 / sR are dead here: team_tables ends with a barrier behind their last readers, and its first barrier holds back
        // every write to the X / Z tables until the slowest wave has finished phase 4)
        for (int k = tl; k < 2 * nEab; k += TEAM) sE12[k] = gEab[(size_t)pab * 2 * nEab + k];
        for (int pcd = 0; pcd < npp_cd; ++pcd) {
            const double q = B.pp_p[cd->pp_off + pcd];
            const double s = p + q, rs = rsqrt(s), pq = p * q, alpha = pq * (rs * rs);
            const double PQ = Pz - B.pp_Pz[cd->pp_off + pcd];
            const double T = alpha * PQ * PQ;
            // 2 pi^(5/2) / (p q sqrt(p+q)) * coefficient product, pyx:1219-1221
            const double pref = Kab * B.pp_K[cd->pp_off + pcd] * (34.986836655249725 * rs) / pq;
            for (int k = tl; k < 2 * nEcd; k += TEAM) {
                const double v = gEcd[(size_t)pcd * 2 * nEcd + k];
                sE34[k] = ((k % Lcd1) & 1) ? -v : v;
            }
            for (int k = tl; k < (L + 1) * RS; k += TEAM) sR[k] = 0.0;
            team_tables<LAB, LCD, TEAM>(B, tl, alpha, PQ, T, sE12, nEab, sE34, nEcd, sR, sG, sX, sZ, nTcd, tc.nT, tc.inv_nTcd);
            // ---- phase 4: parity-allowed components of this primitive quartet, accumulated ----
            int base = 0;
            for (int c = 0; c < 4; ++c) {
                const int nA = tc.pA[c + 1] - tc.pA[c], nK = tc.pK[c + 1] - tc.pK[c];
                if (nA == 0 || nK == 0) continue;
                const float invK = tc.invK[c];
                const int4 *offA = sOffA + tc.pA[c], *offK = sOffK + tc.pK[c];
                for (int j = tl; j < nA * nK; j += TEAM) {
                    const int il = small_div(j, invK), kl = j - il * nK;
                    const int4 oa = offA[il], oc = offK[kl];
                    const double *X = reinterpret_cast<const double *>(bX + (oa.x + oc.x));
                    const double *Y = reinterpret_cast<const double *>(bX + (oa.y + oc.y));
                    const double *Z = reinterpret_cast<const double *>(bZ + (oa.z + oc.z));
                    sAcc[base + j] += pref * team_fact_sum<NM>(X, Y, Z);
                }
                base += nA * nK;
            }
        }
    }
    team_sync<TEAM>();
    // ---- phase 5: ket pair transform inside each parity class, stores to the half-transformed slab ----
    int base = 0;
    for (int c = 0; c < 4; ++c) {
        const int nA = tc.pA[c + 1] - tc.pA[c], nK = tc.pK[c + 1] - tc.pK[c], nS = tc.pS[c + 1] - tc.pS[c];
        if (nA == 0 || nK == 0) continue;
        if (nS > 0) {
            const float invS = tc.invS[c];
            const int4 *offA = sOffA + tc.pA[c];
            const double *scA = sScA + tc.pA[c];
            for (int o = tl; o < nA * nS; o += TEAM) {
                const int il = small_div(o, invS), kap = tc.pS[c] + (o - il * nS);
                const int doff = sDoff[kap];
                if (doff < 0) continue;
                const double *Vr = sAcc + base + il * nK;
                double acc = 0.0;
                for (int e = sTp[kap]; e < sTp[kap + 1]; ++e) acc += sTc[e] * Vr[sTk[e]];
                T2[(size_t)(task.row_first + offA[il].w) * (size_t)tc.RLS + doff] = acc * scA[il];
            }
        }
        base += nA * nK;
    }
}

}  // namespace tfk
