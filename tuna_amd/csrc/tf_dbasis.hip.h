// tf_dbasis.hip.h -- device view of the basis (shells, shell pairs, per-pair tables) shared by the translation units that hold
// ERI kernels (tf_device.hip and tf_eri_team.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "tf_internal.h"
#include "tf_layout.hip.h"

namespace tfk {

struct DShell { int L, ncomp, comp_off, cart_off; };
struct DPair {
    int A, B, La, Lb, npp, pp_off, nE, cls;
    long long e_off;
    int nca, ncb, compoff_a, compoff_b, cartoff_a, cartoff_b;
    int outoff_a, outoff_b;      // first OUTPUT AO (spherical, or Cartesian for CARTHARM) of the two shells; set by tf_build_eri
    int tab_off;                 // component-pair tables of this pair in DBasis::ct_* (nca * ncb entries)
    int pcls[5];                 // ct_ord: component pairs ordered by (x, y) parity class; class c is [pcls[c], pcls[c + 1])
};

// What a shell quartet with the angular momenta (La, Lb | Lc, Ld) needs to know, tabulated once per tf_build_eri (index
// ((La * 6 + Lb) * 6 + Lc) * 6 + Ld): sizes of the per-axis factor tables of eri_cfact_kernel, the index words of their entries
// (DBasis::tup) and the primitive quartets per batch the launch of its shell-pair groups has LDS for.
struct LRec { int L, nM, tsize, nT, xz, gsz, lgG, lgX, nb_cap, tupG_off, tupXZ_off, pad; };

struct DBasis {
    const DShell *shells;
    const DPair *pairs;
    const int8_t *c_lx, *c_ly, *c_lz;
    const double *c_scale;
    const double *pp_p, *pp_Pz, *pp_K;
    const double *epool;
    const double *boys;      // [NGRID][NORD]
    // per-L Cartesian->spherical rows (CSR over the components of one shell): row base sphL_base[L], then ptr/idx/val
    const int *sphL_base, *sphL_ptr, *sphL_idx;
    const double *sphL_val;
    // per shell pair, entry f = ca * ncb + cb (DPair::tab_off): index word ((ax (Lb+1) + bx) | (ay (Lb+1) + by) << 8 | (az (Lb+1) + bz)
    // << 16 | x parity << 24 | y parity << 25), normalisation ratio, position word (ca << 8 | cb), parity-class order
    const int *ct_ix, *ct_pos, *ct_ord;
    const double *ct_sc;
    const LRec *lrec;
    const unsigned short *tup;
    // packed layout (set by tf_build_eri): the layout tables and the stride of a slab row (complete-row shape, max_c NP[c] doubles)
    BLayout bl;
    long long RLS;
    // team kernels (tf_eri_team.hip.h; set by tf_build_eri): per ket pair the slab offsets of its output pairs (kq_off[kq_ptr[pair] + kappa],
    // -1: not stored), per pair class the Cartesian -> output transform of its component pairs, parity class by parity class
    const int *kq_ptr, *kq_off, *kt_ptr, *kt_k;
    const double *kt_c;
    const int *tflat;            // flat component / output lists of the class pairs (TClass::flat_off)
};

}  // namespace tfk
