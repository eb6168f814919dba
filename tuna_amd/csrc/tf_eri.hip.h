// tf_eri.hip.h -- ERI generation kernels (contracted Cartesian shell-quartet blocks into the slab buffer).
// Reference: primitive_pair_eri pyx:1142-1221, contraction pyx:1235-1253, driver pyx:1314-1342.
//
// Launches are made per (bra shell-pair class, ket shell-pair class): every workgroup of a launch sees the same
// angular momenta, component counts and contraction depths (QClass), so the LDS carve-out is exact for the class
// (high occupancy for the cheap classes), loop bounds are wave-uniform, and the Hermite expansion tables of the two
// shell pairs are staged in LDS once per workgroup.
//   eri_class_kernel : one shell quartet per workgroup; primitive quartets in LDS batches; lane groups split the
//                      primitive quartets when the quartet has < 256 Cartesian components (deep contractions).
//   eri_multi_kernel : uncontracted quartets (one primitive quartet) with <= 128 components: G = 256/ncp ket pairs
//                      per workgroup, one lane per component, so the cheap classes (ss|ss ... ) fill the wave.
#pragma once
#include <hip/hip_runtime.h>
#include "tf_kernels.hip.h"

namespace tfk {

struct QClass {
    int La, Lb, Lc, Ld, L, tsize;
    int nca, ncb, ncc, ncd, ncomp;
    int npp_ab, npp_cd, npq;   // maxima over the launch (LDS capacity); the actual depths come from the pair records
    int nEab, nEcd;
    int PB, stride;        // primitive quartets per LDS batch, LDS column stride of the R tables
    int G, ncp;            // multi kernel: shell quartets per workgroup, padded components per quartet
    int n_ket;             // ket pairs in this launch
    int fused, spherical, nsc, nsd, Nout, ld, offBlk;   // fused ket transform: output dims of shells C, D; T2 geometry; LDS block buffer
    int offG;              // factorised kernel: ket half of the z tables, (Lc+1)(Ld+1)(La+Lb+1) nM doubles
    int offTab;            // factorised kernel: per-pair component tables, (nab + ncc*ncd) * 2 doubles (scale, two offset words)
    int offCsr;            // fused spherical ket transform: LDS copy of the two shells' Cartesian->spherical CSR rows (TF_CSR_DOUBLES)
    int tri;               // packed layout: only kets with first shell <= the bra's first shell are needed ((kl) <= (ij))
    int tupG_off, tupXZ_off;   // factorised kernel: index words of its table entries in DBasis::tup (LRec of the class)
    // LDS carve-out, offsets in doubles
    int offR, offPref, offPQ, offRed, offEab, offEcd, offScale, offLmn, lds_doubles;
};

// component info staged in LDS: packed (lx | ly << 8 | lz << 16) and the normalisation ratio
__device__ __forceinline__ void stage_components(const DBasis &B, int comp_off, int ncomp, int *dstLmn, double *dstScale, int tid,
                                                 int nthreads)
{
    for (int c = tid; c < ncomp; c += nthreads) {
        const int i = comp_off + c;
        dstLmn[c] = (int)B.c_lx[i] | ((int)B.c_ly[i] << 8) | ((int)B.c_lz[i] << 16);
        dstScale[c] = B.c_scale[i];
    }
}

struct CompQuartet {
    int lx12, ly12, lz12, lx34, ly34, lz34;
    int ixab, iyab, izab, ixcd, iycd, izcd;
    int ca, cb, cc, cd;
    bool nonzero;
    double cscale;
};

__device__ __forceinline__ void decode_component(const QClass &qc, int c, const int *lmnA, const int *lmnB, const int *lmnC,
                                                 const int *lmnD, const double *scA, const double *scB, const double *scC,
                                                 const double *scD, CompQuartet &Q)
{
    Q.cd = c % qc.ncd; c /= qc.ncd;
    Q.cc = c % qc.ncc; c /= qc.ncc;
    Q.cb = c % qc.ncb; Q.ca = c / qc.ncb;
    const int a = lmnA[Q.ca], b = lmnB[Q.cb], cc = lmnC[Q.cc], d = lmnD[Q.cd];
    const int ax = a & 255, ay = (a >> 8) & 255, az = (a >> 16) & 255;
    const int bx = b & 255, by = (b >> 8) & 255, bz = (b >> 16) & 255;
    const int cx = cc & 255, cy = (cc >> 8) & 255, cz = (cc >> 16) & 255;
    const int dx = d & 255, dy = (d >> 8) & 255, dz = (d >> 16) & 255;
    const int Lab1 = qc.La + qc.Lb + 1, Lcd1 = qc.Lc + qc.Ld + 1;
    Q.lx12 = ax + bx; Q.ly12 = ay + by; Q.lz12 = az + bz;
    Q.lx34 = cx + dx; Q.ly34 = cy + dy; Q.lz34 = cz + dz;
    Q.ixab = (ax * (qc.Lb + 1) + bx) * Lab1; Q.iyab = (ay * (qc.Lb + 1) + by) * Lab1; Q.izab = (az * (qc.Lb + 1) + bz) * Lab1;
    Q.ixcd = (cx * (qc.Ld + 1) + dx) * Lcd1; Q.iycd = (cy * (qc.Ld + 1) + dy) * Lcd1; Q.izcd = (cz * (qc.Ld + 1) + dz) * Lcd1;
    Q.nonzero = !(((Q.lx12 + Q.lx34) & 1) || ((Q.ly12 + Q.ly34) & 1));          // x/y parity, pyx:1324-1327
    Q.cscale = scA[Q.ca] * scB[Q.cb] * scC[Q.cc] * scD[Q.cd];
    if ((Q.lx34 + Q.ly34) & 1) Q.cscale = -Q.cscale;                              // (-1)^(tau+nu) is fixed by parity
}

// The reference's 6-deep Hermite sum (pyx:1179-1217) for one component quartet and one primitive quartet.
// Exy/Ez: tables of the primitive pair ([x|y shared][z]); Rq: R table column (element idx at Rq[idx * stride]).
__device__ __forceinline__ double hermite_sum(const CompQuartet &Q, const double *__restrict__ Exy12, const double *__restrict__ Ez12,
                                              const double *__restrict__ Exy34, const double *__restrict__ Ez34,
                                              const double *__restrict__ Rq, int stride, int L)
{
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
    return sum;
}


#define TF_BLK_DOUBLES 1024   // LDS doubles of the Cartesian (cc,cd) sub-block buffer of the fused ket transform

// LDS copy of the Cartesian->spherical rows of shell types C and D (the global CSR is a chain of dependent L2 loads per output)
#define TF_CSR_CAP 144                                     // entries per shell type (h shells: 11 rows)
#define TF_CSR_DOUBLES (2 * TF_CSR_CAP + (2 * TF_CSR_CAP + 32) / 2)
struct KetCsr { const double *valC, *valD; const int *ptrC, *ptrD, *idxC, *idxD; bool ok; };

__device__ __forceinline__ KetCsr stage_ket_csr(const DBasis &B, const QClass &qc, double *sCsr, int tid, int nthreads)
{
    double *vC = sCsr, *vD = sCsr + TF_CSR_CAP;
    int *ints = reinterpret_cast<int *>(sCsr + 2 * TF_CSR_CAP);
    int *pC = ints, *pD = ints + 16, *iC = ints + 32, *iD = ints + 32 + TF_CSR_CAP;
    const int baseC = B.sphL_base[qc.Lc], baseD = B.sphL_base[qc.Ld];
    const int c0 = B.sphL_ptr[baseC], d0 = B.sphL_ptr[baseD];
    const int nC = B.sphL_ptr[baseC + qc.nsc] - c0, nD = B.sphL_ptr[baseD + qc.nsd] - d0;
    KetCsr K{vC, vD, pC, pD, iC, iD, nC <= TF_CSR_CAP && nD <= TF_CSR_CAP && qc.nsc < 16 && qc.nsd < 16};
    if (!K.ok) return K;
    for (int k = tid; k <= qc.nsc; k += nthreads) pC[k] = B.sphL_ptr[baseC + k] - c0;
    for (int k = tid; k <= qc.nsd; k += nthreads) pD[k] = B.sphL_ptr[baseD + k] - d0;
    for (int k = tid; k < nC; k += nthreads) { iC[k] = B.sphL_idx[c0 + k]; vC[k] = B.sphL_val[c0 + k]; }
    for (int k = tid; k < nD; k += nthreads) { iD[k] = B.sphL_idx[d0 + k]; vD[k] = B.sphL_val[d0 + k]; }
    return K;                                              // the caller's next barrier publishes it
}

// Fused ket transform (reference: the ket half of transform_to_spherical_harmonics, kernel:504-523): the Cartesian values of
// `nblk` complete (cc,cd) sub-blocks sit in LDS (sBlk[b][cc][cd]); every thread produces outputs (b, sc, sd) as short CSR dot
// products and writes them straight into the half-transformed slab T2[row(ca,cb)][k_out][l_out] (and, for the rows layout, the
// mirror image).
__device__ __forceinline__ void ket_epilogue(const DBasis &B, const QClass &qc, const KetCsr &K, const double *sBlk, int nblk,
                                             long long row_first, const DPair &ab, int iab0, const DPair &cd, double *__restrict__ T2,
                                             int tid, int nthreads)
{
    const int per = qc.nsc * qc.nsd;
    const size_t row_len = (size_t)qc.Nout * qc.ld;
    const int baseC = B.sphL_base[qc.Lc], baseD = B.sphL_base[qc.Ld];
    for (int e = tid; e < nblk * per; e += nthreads) {
        const int b = e / per, r = e - b * per;
        const int sc = r / qc.nsd, sd = r - sc * qc.nsd;
        const int k = cd.outoff_a + sc, l = cd.outoff_b + sd;
        size_t dst_index;
        if (qc.tri) {
            // packed layout: slab row (ca, cb) keeps the pairs (k >= l) of its own x/y parity class, at their pair index in the
            // complete-row shape (tf_jkpacked.hip.h); everything else is an exact zero (pyx:1324-1327) or a mirror image
            if (l > k) continue;
            const int wk = B.bl.ao[k], wl = B.bl.ao[l];
            const int cb = (B.ct_ix[ab.tab_off + iab0 + b] >> 24) & 3;
            if ((ao_cls(wk) ^ ao_cls(wl)) != cb) continue;
            dst_index = (size_t)(row_first + b) * (size_t)B.RLS + bl_fullsec(B.bl, cb, ao_cls(wk)) + B.bl.kinfo[(size_t)cb * B.bl.N + ao_sigma(B.bl, wk)].offA +
                        ao_loc(wl);
        } else
            dst_index = (size_t)(row_first + b) * row_len + (size_t)k * qc.ld + l;
        const double *blk = sBlk + (size_t)b * qc.ncc * qc.ncd;
        double s = 0.0;
        if (qc.spherical && K.ok) {
            for (int qa = K.ptrC[sc]; qa < K.ptrC[sc + 1]; ++qa) {
                const double *rowp = blk + K.idxC[qa] * qc.ncd;
                double t = 0.0;
                for (int qb = K.ptrD[sd]; qb < K.ptrD[sd + 1]; ++qb) t += K.valD[qb] * rowp[K.idxD[qb]];
                s += K.valC[qa] * t;
            }
        } else if (qc.spherical) {
            for (int qa = B.sphL_ptr[baseC + sc]; qa < B.sphL_ptr[baseC + sc + 1]; ++qa) {
                const double *rowp = blk + B.sphL_idx[qa] * qc.ncd;
                double t = 0.0;
                for (int qb = B.sphL_ptr[baseD + sd]; qb < B.sphL_ptr[baseD + sd + 1]; ++qb) t += B.sphL_val[qb] * rowp[B.sphL_idx[qb]];
                s += B.sphL_val[qa] * t;
            }
        } else
            s = blk[sc * qc.ncd + sd];
        T2[dst_index] = s;
        if (!qc.tri && cd.A != cd.B) T2[(size_t)(row_first + b) * row_len + (size_t)l * qc.ld + k] = s;   // rows layout: the mirror image
    }
}

// Cooperative Boys/R tables: entry e of the batch (column e of sR) belongs to primitive quartet (pab[e], pcd[e]) of the
// pairs (ab[e], cd[e]).  Lanes e*(L+1)+n, n = 0..L; one barrier per table row.  All threads of the block must call.
template <bool DENSE, class PairOf>
__device__ __forceinline__ void coop_tables_en(const DBasis &B, int L, int n_entries, int stride, double *sR, double *sPref, double *sPQ,
                                               PairOf pair_of, int e, int n, int tid, bool with_ket_weight = true, bool with_bra_weight = true);
template <class PairOf>
__device__ __forceinline__ void coop_tables(const DBasis &B, int L, int n_entries, int stride, double *sR, double *sPref, double *sPQ,
                                            PairOf pair_of, int tid)
{
    const int L1 = L + 1;
    coop_tables_en<false>(B, L, n_entries, stride, sR, sPref, sPQ, pair_of, tid / L1, tid % L1, tid);
}
// the same with the lane's (entry, row position) = (tid / (L + 1), tid % (L + 1)) given (callers in a loop divide once).
// DENSE: the Boys values of entry e are computed by thread e (the first waves, all lanes busy) instead of the lane (e, 0) of the
// row layout -- one wave instead of L + 1 runs the expensive part of the build.
template <bool DENSE, class PairOf>
__device__ __forceinline__ void coop_tables_en(const DBasis &B, int L, int n_entries, int stride, double *sR, double *sPref, double *sPQ,
                                               PairOf pair_of, int e, int n, int tid, bool with_ket_weight, bool with_bra_weight)
{
    const bool mine = e < n_entries;
    if (DENSE ? tid < n_entries : (mine && n == 0)) {
        const int le = DENSE ? tid : e;
        int ppab, ppcd;            // absolute primitive-pair indices
        pair_of(le, ppab, ppcd);
        const double p = B.pp_p[ppab], q = B.pp_p[ppcd];
        const double s = p + q, alpha = p * q / s;
        const double PQ = B.pp_Pz[ppab] - B.pp_Pz[ppcd];
        build_R_row0(sR, stride, le, L, alpha, PQ, B.boys);
        sPQ[le] = PQ;
        // 2 pi^(5/2) / (p q sqrt(p+q)) * coefficient product, pyx:1219-1221
        sPref[le] = (with_bra_weight ? B.pp_K[ppab] : 1.0) * (with_ket_weight ? B.pp_K[ppcd] : 1.0) * (34.986836655249725 / (p * q * sqrt(s)));
    }
    for (int v = 1; v <= L; ++v) {
        __syncthreads();
        if (mine && n <= L - v) {
            const int r0 = tri_index(v, 0, L), r1 = tri_index(v - 1, 0, L);
            double val = sPQ[e] * sR[(r1 + n + 1) * stride + e];
            if (v > 1) val += (double)(v - 1) * sR[(tri_index(v - 2, 0, L) + n + 1) * stride + e];
            sR[(r0 + n) * stride + e] = val;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// GENERIC = true: one launch mixes every class (small problems, where per-class launches cannot fill the GPU); the
// workgroup derives its own class record from the two pair records, LDS offsets are the launch-wide capacities in `qc_in`
// (offEcd - offEab doubles for the bra tables, capE_cd for the ket tables, offPref - offR for the R tables).
template <bool STAGE_E, bool GENERIC>
__global__ __launch_bounds__(TF_ERI_THREADS) void eri_class_kernel(DBasis B, QClass qc_in, const int *__restrict__ bra_pairs,
                                                                   const long long *__restrict__ bra_rowoff,
                                                                   const int *__restrict__ ket_pairs, int Nc,
                                                                   double *__restrict__ Cslab)
{
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const DPair ab = B.pairs[bra_pairs[blockIdx.y]];
    const DPair cd = B.pairs[ket_pairs[blockIdx.x]];
    if (qc_in.tri && cd.A > ab.A) return;                      // every (kl) of this ket lies above every (ij) of the bra
    QClass qc = qc_in;
    bool stage_ok = true;
    if (GENERIC) {
        qc.La = ab.La; qc.Lb = ab.Lb; qc.Lc = cd.La; qc.Ld = cd.Lb;
        qc.L = qc.La + qc.Lb + qc.Lc + qc.Ld;
        qc.tsize = (qc.L + 1) * (qc.L + 2) / 2;
        qc.nca = ab.nca; qc.ncb = ab.ncb; qc.ncc = cd.nca; qc.ncd = cd.ncb;
        qc.ncomp = qc.nca * qc.ncb * qc.ncc * qc.ncd;
        qc.nEab = ab.nE; qc.nEcd = cd.nE;
        int pb = (qc_in.offPref - qc_in.offR) / qc.tsize - 1;
        pb = max(1, min(min(pb, TF_ERI_THREADS), ab.npp * cd.npp));
        qc.PB = pb; qc.stride = pb | 1;
        stage_ok = (ab.npp * 2 * ab.nE <= qc_in.offEcd - qc_in.offEab) && (cd.npp * 2 * cd.nE <= qc_in.offScale - qc_in.offEcd);
    }
    double *sR = smem + qc.offR, *sPref = smem + qc.offPref, *sPQ = smem + qc.offPQ, *sRed = smem + qc.offRed;
    double *sEab = smem + qc.offEab, *sEcd = smem + qc.offEcd, *sScale = smem + qc.offScale;
    int *sLmn = reinterpret_cast<int *>(smem + qc.offLmn);
    const int L = qc.L, stride = qc.stride, PB = qc.PB, ncomp = qc.ncomp;
    const int npp_ab = ab.npp, npp_cd = cd.npp, npq = npp_ab * npp_cd;
    const int nEab = qc.nEab, nEcd = qc.nEcd;
    const long long row0 = bra_rowoff[blockIdx.y];
    const size_t NcNc = (size_t)Nc * Nc;

    // ---- stage component tables and (if they fit) the Hermite expansion tables of both shell pairs ----
    stage_components(B, ab.compoff_a, qc.nca, sLmn, sScale, tid, TF_ERI_THREADS);
    stage_components(B, ab.compoff_b, qc.ncb, sLmn + 21, sScale + 21, tid, TF_ERI_THREADS);
    stage_components(B, cd.compoff_a, qc.ncc, sLmn + 42, sScale + 42, tid, TF_ERI_THREADS);
    stage_components(B, cd.compoff_b, qc.ncd, sLmn + 63, sScale + 63, tid, TF_ERI_THREADS);
    KetCsr kcsr{};
    if (qc.fused && qc.spherical) kcsr = stage_ket_csr(B, qc, smem + qc.offCsr, tid, TF_ERI_THREADS);
    const double *__restrict__ gEab = B.epool + ab.e_off;
    const double *__restrict__ gEcd = B.epool + cd.e_off;
    const bool staged = STAGE_E && stage_ok;
    if (staged) {
        for (int k = tid; k < npp_ab * 2 * nEab; k += TF_ERI_THREADS) sEab[k] = gEab[k];
        for (int k = tid; k < npp_cd * 2 * nEcd; k += TF_ERI_THREADS) sEcd[k] = gEcd[k];
    }
    const double *Eab0 = (GENERIC ? staged : STAGE_E) ? sEab : gEab;
    const double *Ecd0 = (GENERIC ? staged : STAGE_E) ? sEcd : gEcd;
    __syncthreads();

    auto phase1 = [&](int b0, int nb) {
        if (nb * (L + 1) <= TF_ERI_THREADS) {
            coop_tables(B, L, nb, stride, sR, sPref, sPQ,
                        [&](int e, int &ppab, int &ppcd) {
                            const int pq = b0 + e;
                            const int pab = pq / npp_cd;
                            ppab = ab.pp_off + pab;
                            ppcd = cd.pp_off + (pq - pab * npp_cd);
                        }, tid);
        } else if (tid < nb) {
            const int pq = b0 + tid;
            const int pab = pq / npp_cd, pcd = pq - pab * npp_cd;
            const double p = B.pp_p[ab.pp_off + pab], q = B.pp_p[cd.pp_off + pcd];
            const double s = p + q, alpha = p * q / s;
            const double PQ = B.pp_Pz[ab.pp_off + pab] - B.pp_Pz[cd.pp_off + pcd];
            build_R_column(sR, stride, tid, L, alpha, PQ, B.boys);
            sPref[tid] = B.pp_K[ab.pp_off + pab] * B.pp_K[cd.pp_off + pcd] * (34.986836655249725 / (p * q * sqrt(s)));
        }
    };
    auto phase2 = [&](const CompQuartet &Q, int b0, int nb, int g, int NG) -> double {
        double acc = 0.0;
        for (int qq = g; qq < nb; qq += NG) {
            const int pq = b0 + qq;
            const int pab = pq / npp_cd, pcd = pq - pab * npp_cd;
            const double *Exy12 = Eab0 + (size_t)pab * 2 * nEab;
            const double *Exy34 = Ecd0 + (size_t)pcd * 2 * nEcd;
            acc += sPref[qq] * hermite_sum(Q, Exy12, Exy12 + nEab, Exy34, Exy34 + nEcd, sR + qq, stride, L);
        }
        return acc;
    };

    const bool one_batch = npq <= PB;
    if (one_batch) {
        phase1(0, npq);
        __syncthreads();
    }
    // groups of complete (cc,cd) sub-blocks (so that the fused ket transform sees whole blocks); 256-component chunks inside
    double *sBlk = smem + qc.offBlk;
    const int nsub = qc.ncc * qc.ncd, nab = qc.nca * qc.ncb;
    const int GB = qc.fused ? max(1, min(nab, TF_BLK_DOUBLES / nsub)) : nab;
    for (int blk0 = 0; blk0 < nab; blk0 += GB) {
        const int nblk = min(GB, nab - blk0), ncg = nblk * nsub;
        for (int chunk0 = 0; chunk0 < ncg; chunk0 += TF_ERI_THREADS) {
            const int nchunk = min(TF_ERI_THREADS, ncg - chunk0);
            int ncp = 1;
            while (ncp < nchunk) ncp <<= 1;
            const int NG = TF_ERI_THREADS / ncp;                 // lane groups split the primitive quartets of a batch
            const int g = tid / ncp, c0 = tid - g * ncp;
            const bool active = c0 < nchunk;
            CompQuartet Q;
            Q.nonzero = false; Q.cscale = 0.0; Q.ca = Q.cb = Q.cc = Q.cd = 0;
            if (active)
                decode_component(qc, blk0 * nsub + chunk0 + c0, sLmn, sLmn + 21, sLmn + 42, sLmn + 63, sScale, sScale + 21, sScale + 42,
                                 sScale + 63, Q);
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
            if (NG > 1) {                                          // combine the groups in fixed order (reproducible)
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
                if (qc.fused)
                    sBlk[chunk0 + c0] = val;
                else {
                    const size_t row = (size_t)(row0 + (long long)Q.ca * qc.ncb + Q.cb);
                    const int k = cd.cartoff_a + Q.cc, l = cd.cartoff_b + Q.cd;
                    Cslab[row * NcNc + (size_t)k * Nc + l] = val;
                    if (cd.A != cd.B && !qc.tri) Cslab[row * NcNc + (size_t)l * Nc + k] = val;
                }
            }
        }
        if (qc.fused) {
            __syncthreads();
            ket_epilogue(B, qc, kcsr, sBlk, nblk, row0 + blk0, ab, blk0, cd, Cslab, tid, TF_ERI_THREADS);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Uncontracted quartets (npq == 1) with few components: G ket pairs per workgroup, sub-quartet s = tid / ncp.
// LDS: shared bra tables (E, components) + per sub-quartet: R table column, prefactor, ket E tables, ket components.
__global__ __launch_bounds__(TF_ERI_THREADS) void eri_multi_kernel(DBasis B, QClass qc, const int *__restrict__ bra_pairs,
                                                                   const long long *__restrict__ bra_rowoff,
                                                                   const int *__restrict__ ket_pairs, int Nc,
                                                                   double *__restrict__ Cslab)
{
    extern __shared__ double smem[];
    double *sR = smem + qc.offR, *sPref = smem + qc.offPref, *sPQ = smem + qc.offPQ;
    double *sEab = smem + qc.offEab, *sEcd = smem + qc.offEcd, *sScale = smem + qc.offScale;
    int *sLmn = reinterpret_cast<int *>(smem + qc.offLmn);
    const int kc = qc.ncc + qc.ncd;                           // ket component slots per sub-quartet (C then D)
    int *sKet = sLmn + 42 + kc * qc.G;                        // pair id of each sub-quartet (or -1)

    const int tid = threadIdx.x;
    const int G = qc.G, ncp = qc.ncp, L = qc.L, stride = qc.stride;
    const int nEab = qc.nEab, nEcd = qc.nEcd;
    const DPair ab = B.pairs[bra_pairs[blockIdx.y]];
    const long long row0 = bra_rowoff[blockIdx.y];
    const size_t NcNc = (size_t)Nc * Nc;
    const int ket0 = blockIdx.x * G;
    int nsub = min(G, qc.n_ket - ket0);
    if (qc.tri) {                                              // ket lists ascend in the first shell: the needed ones are a prefix
        const int ok = (tid < nsub) && (B.pairs[ket_pairs[ket0 + tid]].A <= ab.A);
        nsub = __syncthreads_count(ok);
        if (nsub == 0) return;
    }

    // ---- staging: bra tables once, ket tables per sub-quartet ----
    stage_components(B, ab.compoff_a, qc.nca, sLmn, sScale, tid, TF_ERI_THREADS);
    stage_components(B, ab.compoff_b, qc.ncb, sLmn + 21, sScale + 21, tid, TF_ERI_THREADS);
    KetCsr kcsr{};
    if (qc.fused && qc.spherical) kcsr = stage_ket_csr(B, qc, smem + qc.offCsr, tid, TF_ERI_THREADS);
    {
        const double *__restrict__ gEab = B.epool + ab.e_off;
        for (int k = tid; k < 2 * nEab; k += TF_ERI_THREADS) sEab[k] = gEab[k];
    }
    for (int s = tid; s < G; s += TF_ERI_THREADS) sKet[s] = (s < nsub) ? ket_pairs[ket0 + s] : -1;
    __syncthreads();
    {
        const int per = 2 * nEcd;
        for (int k = tid; k < nsub * per; k += TF_ERI_THREADS) {
            const int s = k / per, o = k - s * per;
            sEcd[k] = B.epool[B.pairs[sKet[s]].e_off + o];
        }
        for (int k = tid; k < nsub * kc; k += TF_ERI_THREADS) {
            const int s = k / kc, o = k - s * kc;
            const DPair cd = B.pairs[sKet[s]];
            const int i = (o < qc.ncc) ? cd.compoff_a + o : cd.compoff_b + (o - qc.ncc);
            sLmn[42 + k] = (int)B.c_lx[i] | ((int)B.c_ly[i] << 8) | ((int)B.c_lz[i] << 16);
            sScale[42 + k] = B.c_scale[i];
        }
    }
    // ---- phase 1: one R table per sub-quartet, (L+1) lanes each ----
    coop_tables(B, L, nsub, stride, sR, sPref, sPQ,
                [&](int e, int &ppab, int &ppcd) {
                    ppab = ab.pp_off;
                    ppcd = B.pairs[sKet[e]].pp_off;
                }, tid);
    __syncthreads();
    // ---- phase 2: one lane per (sub-quartet, component) ----
    const int s = tid / ncp, c0 = tid - s * ncp;
    if (s < nsub && c0 < qc.ncomp) {
        const int *lmnC = sLmn + 42 + kc * s;
        const double *scC = sScale + 42 + kc * s;
        CompQuartet Q;
        decode_component(qc, c0, sLmn, sLmn + 21, lmnC, lmnC + qc.ncc, sScale, sScale + 21, scC, scC + qc.ncc, Q);
        double val = 0.0;
        if (Q.nonzero) {
            const double *Exy34 = sEcd + (size_t)s * 2 * nEcd;
            val = sPref[s] * hermite_sum(Q, sEab, sEab + nEab, Exy34, Exy34 + nEcd, sR + s, stride, L) * Q.cscale;
        }
        if (qc.fused)
            (smem + qc.offBlk)[s * qc.ncomp + c0] = val;
        else {
            const DPair cd = B.pairs[sKet[s]];
            const size_t row = (size_t)(row0 + (long long)Q.ca * qc.ncb + Q.cb);
            const int k = cd.cartoff_a + Q.cc, l = cd.cartoff_b + Q.cd;
            Cslab[row * NcNc + (size_t)k * Nc + l] = val;
            if (cd.A != cd.B && !qc.tri) Cslab[row * NcNc + (size_t)l * Nc + k] = val;
        }
    }
    if (qc.fused) {
        __syncthreads();
        // all (ca,cb) sub-blocks of every sub-quartet: lanes split by sub-quartet so that each sees its own ket pair
        const int nab = qc.nca * qc.ncb, per = nab * qc.nsc * qc.nsd;
        const int lanes = TF_ERI_THREADS / max(1, nsub);
        const int sq = tid / max(1, lanes), lt = tid - sq * max(1, lanes);
        if (lanes > 0 && sq < nsub) {
            const DPair cd = B.pairs[sKet[sq]];
            (void)per;
            ket_epilogue(B, qc, kcsr, smem + qc.offBlk + (size_t)sq * qc.ncomp, nab, row0, ab, 0, cd, Cslab, lt, lanes);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Factorised kernel for uncontracted shell quartets with many components (ff|ff has 10^4).  The reference's 6-deep sum
// separates into per-axis factors that depend only on the four exponents of that axis:
//     X[ax,bx,cx,dx][m]  = sum_{t+tau = 2m} Ex12[t] Ex34[tau] (-1)^tau (2m-1)!!          (x and y share it: same tables)
//     Z[az,bz,cz,dz][n]  = sum_{v,phi} Ez12[v] Ez34[phi] (-1)^phi R[v+phi][n]
//     (ab|cd) = pref * sum_{m,m'} X[x-tuple][m] X[y-tuple][m'] Z[z-tuple][m+m']
// There are only (La+1)(Lb+1)(Lc+1)(Ld+1) tuples (256 for ffff), so the tables are built once per shell quartet in LDS and
// every Cartesian component costs a handful of multiply-adds instead of ~150 (same terms, different association order:
// results agree with the reference's order to rounding).
// sum_{m + m' <= NM-1} X[m] Y[m'] Z[m + m'] with the three table rows in registers (rows are zero-padded beyond their degree)
template <int NM>
__device__ __forceinline__ double fact_sum(const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z)
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

__device__ __forceinline__ double fact_sum_any(int nM, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z)
{
    switch (nM) {
    case 1: return fact_sum<1>(X, Y, Z);
    case 2: return fact_sum<2>(X, Y, Z);
    case 3: return fact_sum<3>(X, Y, Z);
    case 4: return fact_sum<4>(X, Y, Z);
    case 5: return fact_sum<5>(X, Y, Z);
    case 6: return fact_sum<6>(X, Y, Z);
    case 7: return fact_sum<7>(X, Y, Z);
    case 8: return fact_sum<8>(X, Y, Z);
    case 9: return fact_sum<9>(X, Y, Z);
    default: {
        double sum = 0.0;
        for (int m = 0; m < nM; ++m) {
            double t = 0.0;
            for (int mp = 0; mp + m < nM; ++mp) t += Y[mp] * Z[m + mp];
            sum += X[m] * t;
        }
        return sum;
    }
    }
}

// the same for the contracted kernel, whose accumulators leave fewer registers: unrolled up to 6 rows, a plain loop beyond
__device__ __forceinline__ double fact_sum_small(int nM, const double *__restrict__ X, const double *__restrict__ Y, const double *__restrict__ Z)
{
    switch (nM) {
    case 1: return fact_sum<1>(X, Y, Z);
    case 2: return fact_sum<2>(X, Y, Z);
    case 3: return fact_sum<3>(X, Y, Z);
    case 4: return fact_sum<4>(X, Y, Z);
    case 5: return fact_sum<5>(X, Y, Z);
    case 6: return fact_sum<6>(X, Y, Z);
    default: {
        double sum = 0.0;
        for (int m = 0; m < nM; ++m) {
            double t = 0.0;
            for (int mp = 0; mp + m < nM; ++mp) t += Y[mp] * Z[m + mp];
            sum += X[m] * t;
        }
        return sum;
    }
    }
}

__global__ __launch_bounds__(TF_ERI_THREADS) void eri_fact_kernel(DBasis B, QClass qc, const int *__restrict__ bra_pairs,
                                                                  const long long *__restrict__ bra_rowoff,
                                                                  const int *__restrict__ ket_pairs, int Nc,
                                                                  double *__restrict__ Cslab)
{
    extern __shared__ double smem[];
    double *sR = smem + qc.offR, *sPref = smem + qc.offPref, *sPQ = smem + qc.offPQ;
    double *sEab = smem + qc.offEab, *sEcd = smem + qc.offEcd, *sScale = smem + qc.offScale;
    double *sX = smem + qc.offRed;                               // [nT][nM]
    int *sLmn = reinterpret_cast<int *>(smem + qc.offLmn);
    const int tid = threadIdx.x, nthr = blockDim.x;     // 64, 128 or 256 threads: small classes run one wave per quartet (host: fact_threads)
    const DPair ab = B.pairs[bra_pairs[blockIdx.y]];
    const DPair cd = B.pairs[ket_pairs[blockIdx.x]];
    if (qc.tri && cd.A > ab.A) return;
    const int L = qc.L, nEab = qc.nEab, nEcd = qc.nEcd;
    const int La1 = qc.La + 1, Lb1 = qc.Lb + 1, Lc1 = qc.Lc + 1, Ld1 = qc.Ld + 1;
    const int nT = La1 * Lb1 * Lc1 * Ld1, nM = L / 2 + 1;
    double *sZ = sX + nT * nM;                                   // [nT][nM]
    const int Lab1 = qc.La + qc.Lb + 1, Lcd1 = qc.Lc + qc.Ld + 1;
    const long long row0 = bra_rowoff[blockIdx.y];
    const size_t NcNc = (size_t)Nc * Nc;

    {
        const double *__restrict__ gEab = B.epool + ab.e_off;
        const double *__restrict__ gEcd = B.epool + cd.e_off;
        for (int k = tid; k < 2 * nEab; k += nthr) sEab[k] = gEab[k];
        for (int k = tid; k < 2 * nEcd; k += nthr) sEcd[k] = gEcd[k];
    }
    coop_tables(B, L, 1, 1, sR, sPref, sPQ, [&](int, int &ppab, int &ppcd) { ppab = ab.pp_off; ppcd = cd.pp_off; }, tid);
    __syncthreads();
    // ---- ket half of the z tables: G[c,d][v][n], v <= La + Lb, n <= L - v - (c + d) (zero beyond) ----
    double *sG = smem + qc.offG;
    const unsigned short *__restrict__ tupG = B.tup + qc.tupG_off, *__restrict__ tupXZ = B.tup + qc.tupXZ_off;   // entry index words (host)
    for (int e = tid; e < Lc1 * Ld1 * Lab1 * nM; e += nthr) {
        const int w = tupG[e];
        const int n = w & 15, v = (w >> 4) & 15, d = (w >> 8) & 7, c = (w >> 11) & 7;
        const int l34 = c + d;
        double gsum = 0.0;
        if (n <= L - v - l34) {
            const double *Ez34 = sEcd + nEcd + (c * Ld1 + d) * Lcd1;
            for (int phi = 0; phi <= l34; ++phi) {
                const double term = Ez34[phi] * sR[tri_index(v + phi, n, L)];
                gsum += (phi & 1) ? -term : term;
            }
        }
        sG[e] = gsum;
    }
    __syncthreads();
    // ---- per-axis tables ----
    for (int e = tid; e < nT * nM; e += nthr) {
        const int w = tupXZ[e];
        const int m = w & 15, a = (w >> 4) & 7, b = (w >> 7) & 7, c = (w >> 10) & 7, d = (w >> 13) & 7;
        const int l12 = a + b, l34 = c + d;
        const double *E12 = sEab + (a * Lb1 + b) * Lab1, *E34 = sEcd + (c * Ld1 + d) * Lcd1;
        // X: t + tau = 2m, with t = l12 (mod 2) and tau = l34 (mod 2) -- other parities have zero coefficients
        double x = 0.0;
        if (((l12 + l34) & 1) == 0 && 2 * m <= l12 + l34) {
            for (int t = l12 & 1; t <= l12; t += 2) {
                const int tau = 2 * m - t;
                if (tau < 0 || tau > l34) continue;
                const double term = E12[t] * E34[tau];
                x += (tau & 1) ? -term : term;
            }
            x *= c_dfact[m];
        }
        sX[e] = x;
        // Z in two stages (below): first the ket half G[c,d][v][n] = sum_phi (-1)^phi Ez34[phi] R[v + phi][n], then the bra half
        double z = 0.0;
        if (m <= L - l12 - l34) {
            const double *Ez12 = sEab + nEab + (a * Lb1 + b) * Lab1;
            const double *G = sG + ((c * Ld1 + d) * Lab1) * nM + m;
            for (int v = 0; v <= l12; ++v) z += Ez12[v] * G[v * nM];
        }
        sZ[e] = z;
    }
    __syncthreads();
    // the R / E tables are dead from here on: the spherical CSR rows of the ket shells may be staged over them (host: offCsr)
    KetCsr kcsr{};
    if (qc.fused && qc.spherical) kcsr = stage_ket_csr(B, qc, smem + qc.offCsr, tid, nthr);
    // ---- components, in groups of complete (cc,cd) sub-blocks ----
    // Per bra component pair and per ket component pair (class constants, tabulated once per workgroup): the pair's part of the three
    // table indices, the parities of its x and y exponent sums, and its normalisation ratio.  A component is then two table lookups,
    // one parity test and one unrolled triple-table sum -- no integer division by runtime shell sizes, no per-component lmn decode.
    // The tables are staged as BYTE OFFSETS into the X / Z tables (16-bit fields, the class constants (Lc+1)(Ld+1) and nM multiplied in
    // once per entry) and the scale of a bra pair carries the quartet's prefactor: the component loop itself has no integer
    // multiplication (quarter rate on this hardware: nine of them cost as much as the 35 multiply-adds of an (ff|ff) component).
    const double pref = sPref[0];
    double *sBlk = smem + qc.offBlk;
    const int nsubc = qc.ncc * qc.ncd, nab = qc.nca * qc.ncb;
    double *sScAB = smem + qc.offTab, *sScCD = sScAB + nab;
    int2 *sOfAB = reinterpret_cast<int2 *>(sScCD + nsubc), *sOfCD = sOfAB + nab;
    const int LcLd = Lc1 * Ld1;
    for (int e = tid; e < nab + nsubc; e += nthr) {         // host-tabulated per shell pair (DBasis::ct_*)
        const bool bra = e < nab;
        const int f = bra ? e : e - nab, g = (bra ? ab.tab_off : cd.tab_off) + f;
        const int w = B.ct_ix[g], unit = (bra ? LcLd : 1) * nM * (int)sizeof(double);
        const int2 o = make_int2(((w & 255) * unit) | ((((w >> 8) & 255) * unit) << 16), (((w >> 16) & 255) * unit) | (((w >> 24) & 3) << 16));
        if (bra) { sOfAB[f] = o; sScAB[f] = pref * B.ct_sc[g]; }
        else { sOfCD[f] = o; sScCD[f] = B.ct_sc[g]; }
    }
    __syncthreads();
    const char *bX = reinterpret_cast<const char *>(sX), *bZ = reinterpret_cast<const char *>(sZ);
    const int GB = qc.fused ? max(1, min(nab, TF_BLK_DOUBLES / nsubc)) : nab;
    // component cl = abl * nsubc + icd of a block; a thread's components advance by the workgroup size: (abl, icd) incrementally
    const int step_a = nthr / nsubc, step_c = nthr - step_a * nsubc;
    const int abl_first = tid / nsubc, icd_first = tid - abl_first * nsubc;
    for (int blk0 = 0; blk0 < nab; blk0 += GB) {
        const int nblk = min(GB, nab - blk0), ncg = nblk * nsubc;
        int abl = abl_first, icd = icd_first;
        for (int cl = tid; cl < ncg; cl += nthr) {
            const int iab = blk0 + abl;
            const int2 oa = sOfAB[iab], oc = sOfCD[icd];
            double val = 0.0;
            if ((((oa.y ^ oc.y) >> 16) & 3) == 0) {                      // x and y exponent sums both even, pyx:1324-1327
                const double *X = reinterpret_cast<const double *>(bX + ((oa.x & 0xffff) + (oc.x & 0xffff)));
                const double *Y = reinterpret_cast<const double *>(bX + (((unsigned)oa.x >> 16) + ((unsigned)oc.x >> 16)));
                const double *Z = reinterpret_cast<const double *>(bZ + ((oa.y & 0xffff) + (oc.y & 0xffff)));
                val = fact_sum_any(nM, X, Y, Z) * (sScAB[iab] * sScCD[icd]);
            }
            if (qc.fused)
                sBlk[cl] = val;
            else {
                const int ia = iab / qc.ncb, ib = iab - ia * qc.ncb, ic = icd / qc.ncd, id = icd - ic * qc.ncd;
                const size_t row = (size_t)(row0 + (long long)ia * qc.ncb + ib);
                const int k = cd.cartoff_a + ic, l = cd.cartoff_b + id;
                Cslab[row * NcNc + (size_t)k * Nc + l] = val;
                if (cd.A != cd.B && !qc.tri) Cslab[row * NcNc + (size_t)l * Nc + k] = val;
            }
            abl += step_a; icd += step_c;
            if (icd >= nsubc) { icd -= nsubc; ++abl; }
        }
        if (qc.fused) {
            __syncthreads();
            ket_epilogue(B, qc, kcsr, sBlk, nblk, row0 + blk0, ab, blk0, cd, Cslab, tid, nthr);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Factorised kernel for CONTRACTED shell quartets, class-agnostic (one launch mixes every class of a group of shell pairs).
// eri_class_kernel gives every lane one Cartesian component and lets it run the reference's 6-deep Hermite sum per primitive
// quartet: the loop bounds differ from lane to lane, so a wave executes the worst-case nest for every component (measured on
// Ar2/cc-pVQZ: ~2 400 vector instructions per component and primitive quartet, a (pp|pp) quartet of two 8-primitive shells runs
// for 12 ms).  Here the per-axis tables of eri_fact_kernel are built for a BATCH of primitive quartets at a time:
//     X_q[ax,bx,cx,dx][m], Z_q[az,bz,cz,dz][n]   (q = primitive quartet of the batch; definitions above eri_fact_kernel)
//     (ab|cd) = sum_q pref_q * sum_{m,m'} X_q[x-tuple][m] X_q[y-tuple][m'] Z_q[z-tuple][m+m']
// Table entries are spread over the workgroup by (primitive, entry) with precomputed index words (no integer divisions in the
// batch loop); a component costs one unrolled triple-table sum per primitive quartet.  Components < 256: lane groups split the
// primitive quartets of a batch and are combined in fixed order at the end; more: up to TF_CF_KMAX components per thread in
// registers.  Uncontracted quartets (one primitive quartet) take the same path with direct stores.  Cartesian output (unfused).
#define TF_CF_KMAX 2
struct CFCaps {
    int offR, capR, offPref, offPQ, offPP, offG, capG, offX, offZ, capXZ, offTupG, offTupXZ, offEab, capEab, offEcd, capEcd;
    int offRed, lds_doubles, tri;
    int gtab_doubles;                // GTAB launches: offG / offX / offZ are relative to the workgroup's block of this many doubles in global memory
    int dbg_npq_lo, dbg_npq_hi;      // profiling aid (TF_ERI_DBG_NPQ=lo:hi): only quartets with lo <= primitive quartets <= hi are computed
    int offKm, capKm;                // FAMILY launches: weights of the members' primitive pairs, [member][primitive pair] (capKm doubles)
    int team_lmax, team_pqmax;       // quartets with both pair sums <= team_lmax and <= team_pqmax primitive quartets belong to eri_teamc_kernel (-1: none)
};

// GTAB: the G / X / Z tables of the workgroup live in global memory (gtab, one block of cap.gtab_doubles per workgroup, L2-resident)
// instead of LDS -- the very top of the angular momenta ((hh|hh): 2 x 14 256 + 4 356 doubles) does not fit the 160 KB.
// FAMILIES (MM > 1; general contractions, round 3): TUNA keeps every contracted function of a generally contracted set as a shell of its
// own that repeats the whole primitive list (mol:553-574: Ar cc-pVQZ = three s shells on the same 13 primitives, two p shells on the same
// 8), so the ket pairs (C_i, D_j) over shells with identical primitives differ in NOTHING but the weights of their primitive pairs and
// the AOs they write.  A workgroup takes one bra pair and a FAMILY of up to MM such ket pairs (ket_pairs[] lists the family heads;
// members fam_mem[fam_ptr[x] .. fam_ptr[x + 1])): the tables of a primitive quartet and the triple-table sum of a component are
// evaluated once and feed one accumulator per member -- (xx|s13 s13) nine for the price of one.  Same terms as before, the member's
// weight multiplied in last.
template <bool UNC, bool GTAB = false, int MA = 1, int MC = 1>
__global__ __launch_bounds__(TF_ERI_THREADS, MA * MC > 9 ? 2 : (MA * MC > 1 ? 3 : 4)) void eri_cfact_kernel(DBasis B, CFCaps cap, const int *__restrict__ bra_pairs,
                                                                   const long long *__restrict__ bra_rowoff,
                                                                   const int *__restrict__ ket_pairs, int Nc, double *__restrict__ Cslab,
                                                                   double *__restrict__ gtab = nullptr, const int *__restrict__ fam_ptr = nullptr,
                                                                   const int *__restrict__ fam_mem = nullptr, const int *__restrict__ bfam_ptr = nullptr,
                                                                   const int *__restrict__ bfam_mem = nullptr)
{
    // MC > 1: blockIdx.x runs over the heads of ket families (ket_pairs[] = heads, members fam_mem[fam_ptr[x] ..]: pair indices);
    // MA > 1: blockIdx.y over the heads of bra families (bra_pairs[] / bra_rowoff[] = the slab's lists as before; bfam_mem[bfam_ptr[y] ..]:
    // positions in those lists, the first one the head)
    constexpr int MM = MA * MC;
    constexpr int KM = MM > 9 ? 1 : TF_CF_KMAX;                        // components per lane and pass (registers: KM x MM accumulators)
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    int nma = 1, nmc = 1;
    const int *__restrict__ memA = nullptr, *__restrict__ mem = nullptr;
    if (MA > 1) { memA = bfam_mem + bfam_ptr[blockIdx.y]; nma = bfam_ptr[blockIdx.y + 1] - bfam_ptr[blockIdx.y]; }
    const int ybra = MA > 1 ? memA[0] : (int)blockIdx.y;
    const DPair ab = B.pairs[bra_pairs[ybra]];
    const DPair cd = B.pairs[ket_pairs[blockIdx.x]];
    if (MC > 1) { mem = fam_mem + fam_ptr[blockIdx.x]; nmc = fam_ptr[blockIdx.x + 1] - fam_ptr[blockIdx.x]; }
    if (MM > 1) {
        bool any = false;
        for (int ma = 0; ma < nma; ++ma) {
            const int Aab = MA > 1 ? B.pairs[bra_pairs[memA[ma]]].A : ab.A;
            for (int mc = 0; mc < nmc; ++mc) any = any || !(cap.tri && (MC > 1 ? B.pairs[mem[mc]].A : cd.A) > Aab);
        }
        if (!any) return;
    } else if (cap.tri && cd.A > ab.A) return;
    if (ab.npp * cd.npp < cap.dbg_npq_lo || ab.npp * cd.npp > cap.dbg_npq_hi) return;
    if (cap.team_lmax >= 0 && ab.La + ab.Lb <= cap.team_lmax && cd.La + cd.Lb <= cap.team_lmax && ab.npp * cd.npp <= cap.team_pqmax) return;
    const LRec lr = B.lrec[((ab.La * 6 + ab.Lb) * 6 + cd.La) * 6 + cd.Lb];
    const int Lb1 = ab.Lb + 1, Lc1 = cd.La + 1, Ld1 = cd.Lb + 1;
    const int Lab1 = ab.La + ab.Lb + 1, Lcd1 = cd.La + cd.Lb + 1;
    const int L = lr.L, nM = lr.nM, xz = lr.xz, gsz = lr.gsz;
    const int nsubc = cd.nca * cd.ncb;
    const int npp_cd = cd.npp, npq = ab.npp * cd.npp;
    const int nEab = ab.nE, nEcd = cd.nE;
    double *sR = smem + cap.offR, *sPref = smem + cap.offPref, *sPQ = smem + cap.offPQ, *sRed = smem + cap.offRed;
    double *tab = GTAB ? gtab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (size_t)cap.gtab_doubles : smem;
    double *sG = tab + cap.offG, *sX = tab + cap.offX, *sZ = tab + cap.offZ;
    int *sPP = reinterpret_cast<int *>(smem + cap.offPP);            // [2][256]: primitive pair indices (within the pairs) of the batch
    unsigned short *sTupG = reinterpret_cast<unsigned short *>(smem + cap.offTupG), *sTupXZ = reinterpret_cast<unsigned short *>(smem + cap.offTupXZ);
    const long long row0 = bra_rowoff[ybra];
    const size_t NcNc = (size_t)Nc * Nc;
    // primitive quartets per batch: what the cooperative R build and the table capacities of this launch allow (LRec::nb_cap)
    const int NB = min(npq, lr.nb_cap);
    const int stride = NB | 1;

    const double *__restrict__ gEab = B.epool + ab.e_off;
    const double *__restrict__ gEcd = B.epool + cd.e_off;
    const bool stAB = ab.npp * 2 * nEab <= cap.capEab, stCD = cd.npp * 2 * nEcd <= cap.capEcd;
    if (stAB) for (int k = tid; k < ab.npp * 2 * nEab; k += TF_ERI_THREADS) smem[cap.offEab + k] = gEab[k];
    if (stCD) for (int k = tid; k < cd.npp * 2 * nEcd; k += TF_ERI_THREADS) smem[cap.offEcd + k] = gEcd[k];
    const double *Eab0 = stAB ? smem + cap.offEab : gEab;
    const double *Ecd0 = stCD ? smem + cap.offEcd : gEcd;
    // index words of the table entries (tabulated by the host per (La, Lb | Lc, Ld)): G entry r -> (c, d, v, n); X/Z entry r -> (a, b, c, d, m)
    for (int r = tid; r < gsz; r += TF_ERI_THREADS) sTupG[r] = B.tup[lr.tupG_off + r];
    for (int r = tid; r < xz; r += TF_ERI_THREADS) sTupXZ[r] = B.tup[lr.tupXZ_off + r];
    // Component pairs of the two shell pairs (tabulated by the host, DBasis::ct_*): index words, normalisation ratios, positions,
    // and their order by (x, y) parity class.  Only components whose bra and ket classes agree are non-zero (pyx:1324-1327; the
    // slab is zeroed by the host): non-zero component j <-> (class, i-th bra pair, k-th ket pair of the class).
    const int *__restrict__ ixAB = B.ct_ix + ab.tab_off, *__restrict__ ixCD = B.ct_ix + cd.tab_off;
    const int *__restrict__ ordAB = B.ct_ord + ab.tab_off, *__restrict__ ordCD = B.ct_ord + cd.tab_off;
    const double *__restrict__ scAB = B.ct_sc + ab.tab_off, *__restrict__ scCD = B.ct_sc + cd.tab_off;
    const int *__restrict__ posCD = B.ct_pos + cd.tab_off;
    // (scalars, not arrays: a dynamically indexed local array would live in scratch memory)
    const int oa0 = ab.pcls[0], oa1 = ab.pcls[1], oa2 = ab.pcls[2], oa3 = ab.pcls[3], oa4 = ab.pcls[4];
    const int oc0 = cd.pcls[0], oc1 = cd.pcls[1], oc2 = cd.pcls[2], oc3 = cd.pcls[3], oc4 = cd.pcls[4];
    const int pre1 = (oa1 - oa0) * (oc1 - oc0), pre2 = pre1 + (oa2 - oa1) * (oc2 - oc1), pre3 = pre2 + (oa3 - oa2) * (oc3 - oc2);
    const int nnz = pre3 + (oa4 - oa3) * (oc4 - oc3);
    const int LcLd = Lc1 * Ld1;
    __syncthreads();
    // non-zero component j: the two pair indices and the table offsets (X row of the x tuple, X row of the y tuple, Z row)
    auto comp_rows = [&](int j, int &xo, int &yo, int &zo, int &iab, int &icd) {
        const bool g1 = j >= pre1, g2 = j >= pre2, g3 = j >= pre3;
        const int base = g3 ? pre3 : (g2 ? pre2 : (g1 ? pre1 : 0));
        const int offa = g3 ? oa3 : (g2 ? oa2 : (g1 ? oa1 : oa0)), offc = g3 ? oc3 : (g2 ? oc2 : (g1 ? oc1 : oc0));
        const int ncd_c = (g3 ? oc4 : (g2 ? oc3 : (g1 ? oc2 : oc1))) - offc;
        const int r = j - base;
        const int i = r / ncd_c, k = r - i * ncd_c;
        iab = ordAB[offa + i]; icd = ordCD[offc + k];
        const int pa = ixAB[iab], pc = ixCD[icd];
        xo = ((pa & 255) * LcLd + (pc & 255)) * nM;
        yo = (((pa >> 8) & 255) * LcLd + ((pc >> 8) & 255)) * nM;
        zo = (((pa >> 16) & 255) * LcLd + ((pc >> 16) & 255)) * nM;
    };
    auto store_to = [&](const DPair &kp, long long rowbase, int iab, int icd, double val) {
        const int pos = posCD[icd];
        const size_t row = (size_t)(rowbase + iab);
        const int k = kp.cartoff_a + (pos >> 8), l = kp.cartoff_b + (pos & 255);
        Cslab[row * NcNc + (size_t)k * Nc + l] = val;
        if (kp.A != kp.B && !cap.tri) Cslab[row * NcNc + (size_t)l * Nc + k] = val;
    };
    auto store = [&](int iab, int icd, double val) { store_to(cd, row0, iab, icd, val); };
    // FAMILIES: the members' primitive-pair weights, staged once: [ket member][primitive pair of the ket], then [bra member][.. of the bra]
    double *sKc = smem + cap.offKm, *sKa = sKc + (MC > 1 ? nmc * npp_cd : 0);
    const int npp_ab = ab.npp;
    if (MC > 1)
        for (int e = tid; e < nmc * npp_cd; e += TF_ERI_THREADS) { const int mm = e / npp_cd; sKc[e] = B.pp_K[B.pairs[mem[mm]].pp_off + (e - mm * npp_cd)]; }
    if (MA > 1)
        for (int e = tid; e < nma * npp_ab; e += TF_ERI_THREADS) { const int mm = e / npp_ab; sKa[e] = B.pp_K[B.pairs[bra_pairs[memA[mm]]].pp_off + (e - mm * npp_ab)]; }
    // (made visible by the barriers of the first build_tables)

    // tables of the primitive quartets b0 .. b0 + nb - 1
    const int lgG = lr.lgG, lgX = lr.lgX, rstepG = 1 << lgG, rstepX = 1 << lgX;
    const int ce = tid / (L + 1), cn = tid - ce * (L + 1);            // lane (entry, row position) of the cooperative R build
    auto build_tables = [&](int b0, int nb) {
        coop_tables_en<true>(B, L, nb, stride, sR, sPref, sPQ,
                       [&](int e, int &ppab, int &ppcd) {
                           const int pq = b0 + e;
                           const int pab = pq / npp_cd, pcd = pq - pab * npp_cd;
                           sPP[e] = pab; sPP[TF_ERI_THREADS + e] = pcd;
                           ppab = ab.pp_off + pab; ppcd = cd.pp_off + pcd;
                       }, ce, cn, tid, MC == 1, MA == 1);
        __syncthreads();
        // ket half of the z tables: G_q[c,d][v][n] = sum_phi (-1)^phi Ez34[phi] R_q[v + phi][n]
        for (int p0 = 0; p0 < nb; p0 += TF_ERI_THREADS >> lgG) {
            const int q = p0 + (tid >> lgG);
            if (q >= nb) continue;
            const double *Ez34q = Ecd0 + (size_t)sPP[TF_ERI_THREADS + q] * 2 * nEcd + nEcd;
            for (int r = tid & (rstepG - 1); r < gsz; r += rstepG) {
                const int w = sTupG[r];
                const int n = w & 15, v = (w >> 4) & 15, d = (w >> 8) & 7, c = (w >> 11) & 7;
                const int l34 = c + d;
                double gsum = 0.0;
                if (n <= L - v - l34) {
                    const double *Ez34 = Ez34q + (c * Ld1 + d) * Lcd1;
                    for (int phi = 0; phi <= l34; ++phi) {
                        const double term = Ez34[phi] * sR[tri_index(v + phi, n, L) * stride + q];
                        gsum += (phi & 1) ? -term : term;
                    }
                }
                sG[q * gsz + r] = gsum;
            }
        }
        __syncthreads();
        for (int p0 = 0; p0 < nb; p0 += TF_ERI_THREADS >> lgX) {
            const int q = p0 + (tid >> lgX);
            if (q >= nb) continue;
            const double *E12q = Eab0 + (size_t)sPP[q] * 2 * nEab, *E34q = Ecd0 + (size_t)sPP[TF_ERI_THREADS + q] * 2 * nEcd;
            for (int r = tid & (rstepX - 1); r < xz; r += rstepX) {
                const int w = sTupXZ[r];
                const int m = w & 15, a = (w >> 4) & 7, b = (w >> 7) & 7, c = (w >> 10) & 7, d = (w >> 13) & 7;
                const int l12 = a + b, l34 = c + d;
                const double *E12 = E12q + (a * Lb1 + b) * Lab1, *E34 = E34q + (c * Ld1 + d) * Lcd1;
                double x = 0.0;
                if (((l12 + l34) & 1) == 0 && 2 * m <= l12 + l34) {
                    for (int t = l12 & 1; t <= l12; t += 2) {
                        const int tau = 2 * m - t;
                        if (tau < 0 || tau > l34) continue;
                        const double term = E12[t] * E34[tau];
                        x += (tau & 1) ? -term : term;
                    }
                    x *= c_dfact[m];
                }
                sX[q * xz + r] = x;
                double z = 0.0;
                if (m <= L - l12 - l34) {
                    const double *Ez12 = E12 + nEab;
                    const double *G = sG + q * gsz + ((c * Ld1 + d) * Lab1) * nM + m;
                    for (int v = 0; v <= l12; ++v) z += Ez12[v] * G[v * nM];
                }
                sZ[q * xz + r] = z;
            }
        }
        __syncthreads();
    };

    if (UNC) {                                                        // uncontracted (npq == 1 for every quartet of the launch): one batch, direct stores
        build_tables(0, 1);
        const double pref = sPref[0];
        for (int j = tid; j < nnz; j += TF_ERI_THREADS) {
            int xo, yo, zo, iab, icd;
            comp_rows(j, xo, yo, zo, iab, icd);
            store(iab, icd, pref * fact_sum_any(nM, sX + xo, sX + yo, sZ + zo) * (scAB[iab] * scCD[icd]));
        }
        return;
    }
    // fewer than 256 non-zero components: NG lane groups split the primitive quartets of a batch
    const int NG = nnz < TF_ERI_THREADS ? TF_ERI_THREADS / nnz : 1, ncp = NG > 1 ? nnz : TF_ERI_THREADS;
    const int g = tid / ncp, c0 = tid - g * ncp;
    const bool lane_on = g < NG;
    for (int cbase = 0; cbase < nnz; cbase += TF_ERI_THREADS * KM) {
        int xo[KM], yo[KM], zo[KM], iab[KM], icd[KM];
        double acc[KM][MM];
        const int nk = min(KM, (nnz - cbase + TF_ERI_THREADS - 1) / TF_ERI_THREADS);
#pragma unroll
        for (int k = 0; k < KM; ++k) {
#pragma unroll
            for (int mm = 0; mm < MM; ++mm) acc[k][mm] = 0.0;
            xo[k] = -1; yo[k] = zo[k] = iab[k] = icd[k] = 0;
            const int j = cbase + k * TF_ERI_THREADS + c0;
            if (k < nk && j < nnz && lane_on) comp_rows(j, xo[k], yo[k], zo[k], iab[k], icd[k]);
        }
        for (int b0 = 0; b0 < npq; b0 += NB) {
            const int nb = min(NB, npq - b0);
            build_tables(b0, nb);
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                if (k >= nk || xo[k] < 0) continue;
                if (MM == 1) {
                    double a = 0.0;
                    for (int q = g; q < nb; q += NG) a += sPref[q] * fact_sum_small(nM, sX + q * xz + xo[k], sX + q * xz + yo[k], sZ + q * xz + zo[k]);
                    acc[k][0] += a;
                } else {
                    for (int q = g; q < nb; q += NG) {
                        const double f = sPref[q] * fact_sum_small(nM, sX + q * xz + xo[k], sX + q * xz + yo[k], sZ + q * xz + zo[k]);
                        // weights of this primitive quartet's pairs in the members: Kc[mc * npp_cd], Ka[ma * npp_ab]
                        const double *Kc = sKc + sPP[TF_ERI_THREADS + q], *Ka = sKa + sPP[q];
#pragma unroll
                        for (int ma = 0; ma < MA; ++ma) {
                            if (ma >= nma) break;
                            const double fa = MA > 1 ? f * Ka[ma * npp_ab] : f;
#pragma unroll
                            for (int mc = 0; mc < MC; ++mc)
                                if (mc < nmc) acc[k][ma * MC + mc] += MC > 1 ? fa * Kc[mc * npp_cd] : fa;
                        }
                    }
                }
            }
            __syncthreads();                                          // the tables are rebuilt by the next batch
        }
        if (NG > 1) {                                                 // combine the lane groups in fixed order (reproducible)
#pragma unroll
            for (int mm = 0; mm < MM; ++mm) {
                if (mm / MC >= nma) break;
                if (mm % MC >= nmc) continue;                         // (uniform over the workgroup: the barriers below match)
                sRed[tid] = acc[0][mm];
                __syncthreads();
                if (g == 0) {
                    double t = 0.0;
                    for (int gg = 0; gg < NG; ++gg) t += sRed[gg * ncp + c0];
                    acc[0][mm] = t;
                }
                __syncthreads();
            }
        }
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (!(k < nk && xo[k] >= 0 && g == 0)) continue;
            if (MM == 1) store(iab[k], icd[k], acc[k][0] * (scAB[iab[k]] * scCD[icd[k]]));
            else {
#pragma unroll
                for (int ma = 0; ma < MA; ++ma) {
                    if (ma >= nma) break;
                    const int yb = MA > 1 ? memA[ma] : ybra;
                    const int Aab = MA > 1 ? B.pairs[bra_pairs[yb]].A : ab.A;
                    const long long rowb = MA > 1 ? bra_rowoff[yb] : row0;
#pragma unroll
                    for (int mc = 0; mc < MC; ++mc) {
                        if (mc >= nmc) break;
                        const DPair kp = MC > 1 ? B.pairs[mem[mc]] : cd;
                        if (cap.tri && kp.A > Aab) continue;         // (every (kl) of this member lies above every (ij) of that bra)
                        store_to(kp, rowb, iab[k], icd[k], acc[k][ma * MC + mc] * (scAB[iab[k]] * scCD[icd[k]]));
                    }
                }
            }
        }
    }
}

}  // namespace tfk
