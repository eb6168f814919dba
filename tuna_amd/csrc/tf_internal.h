// tf_internal.h -- shared host/device data model of libtunafock (not part of the public ABI).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#define TF_MAX_L 5          // shells up to H (tuna_molecule.py:612-618)
#define TF_BOYS_TMAX 36.0   // Taylor grid covers [0, TMAX]; beyond: erf-asymptote + upward recursion
#define TF_BOYS_STEP 0.125
#define TF_BOYS_NGRID 289   // 0, 1/8, ..., 36
#define TF_BOYS_NORD 30     // orders 0..29 tabulated (top order 4*5 = 20, +8 Taylor terms)

namespace tf {

struct Shell {
    double z;        // centre on the z axis
    int atom;        // 0/1: index of the centre (by distinct z)
    int L;
    int nprim;
    int prim_off;    // into Basis::s_exp / s_w
    int ncomp;       // Cartesian components carried by this shell ((L+1)(L+2)/2 when `full`)
    int comp_off;    // into Basis::c_l* / c_scale
    int cart_off;    // first Cartesian AO
    int sph_off;     // first spherical AO (valid when Basis::all_full)
    int nsph;
    bool full;       // components are the canonical x^L..z^L list
};

// One shell pair A >= B (AO index order) with everything that depends on two shells only
// (reference: AOPairERI / PrimitivePairERI, pyx:35-67, built at pyx:1050-1128).
struct Pair {
    int A, B;
    int La, Lb;
    int npp;             // primitive pairs = nprim_A * nprim_B
    int pp_off;          // into pp_p / pp_Pz / pp_K
    long long e_off;     // into E pool: per primitive pair [Exy (nE) | Ez (nE)], nE = (La+1)(Lb+1)(La+Lb+1)
    int nE;
};

struct Basis {
    int n_cart = 0, n_sph = 0;
    bool all_full = true;
    std::vector<Shell> shells;
    // per-AO copy of what the caller passed + reference-exact normalisation (pyx:174-210)
    std::vector<double> ao_origin;   // 3n
    std::vector<int32_t> ao_lmn;     // 3n
    std::vector<int32_t> ao_prim_off;
    std::vector<double> ao_exp, ao_coef_raw, ao_coef, ao_norm;
    std::vector<int32_t> ao_shell;   // shell of each Cartesian AO
    // shell-level primitives: exponent and weight (= norm*coef of the shell's FIRST component)
    std::vector<double> s_exp, s_w;
    // component table
    std::vector<int8_t> c_lx, c_ly, c_lz;
    std::vector<double> c_scale;     // weight of this component relative to the first one
    // pairs
    std::vector<Pair> pairs;
    std::vector<double> pp_p, pp_Pz, pp_K, pp_AB;
    std::vector<double> epool;
    // AO-level CSR of the Cartesian->spherical map (row = output AO; identity when cartesian output)
    std::vector<int32_t> sph_ptr, sph_idx;
    std::vector<double> sph_val;
};

// pyx:174-210 for one AO
void normalize_ao(int l, int m, int n, int nprim, const double *exps, double *coefs, double *norm);
// pyx:961-1036: full table E[i][j][t], i<=l1, j<=l2, t<=l1+l2, layout ((i*(l2+1)+j)*(l1+l2+1)+t)
void hermite_table(int l1, int l2, double R, double a, double b, double *E);
// Build everything from the flat AO list.  Returns "" or an error message.
std::string build_basis(Basis &bs, int n, const double *origin, const int32_t *lmn, const int32_t *prim_off,
                        const double *exps, const double *coefs_raw);
// Cartesian->spherical block for one L: rows 2L+1, cols (L+1)(L+2)/2, reference row order.
void sph_block(int L, std::vector<double> &U);
// Boys Taylor table F_m(i*STEP), [NGRID][NORD]
void boys_table(std::vector<double> &tab);
// dense U [n_sph, n_cart]
void dense_sph_matrix(const Basis &bs, std::vector<double> &U);

}  // namespace tf
