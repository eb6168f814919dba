// tf_host.cpp -- host-side preparation: AO normalisation, shell grouping, shell-pair Hermite data,
// Cartesian->spherical coefficients, Boys Taylor grid.  Everything here is O(N^2 K^2) set-up that the
// reference does serially under the GIL (pyx:1300-1308); the N^4 work is in tf_device.hip.
#include "tf_internal.h"

#include <cmath>
#include <complex>
#include <cstring>

namespace tf {

static double dfact(int n)  // n!! with (n <= 0) -> 1 (pyx:245-271)
{
    double r = 1.0;
    while (n > 1) { r *= n; n -= 2; }
    return r;
}

// Basis.normalize (pyx:174-210): primitive norms, then the contracted renormalisation in place.
void normalize_ao(int l, int m, int n, int nprim, const double *exps, double *coefs, double *norm)
{
    const double PI = 3.141592653589793238462643383279;
    int L = l + m + n;
    for (int i = 0; i < nprim; ++i)
        norm[i] = std::sqrt(std::pow(2, 2 * L + 1.5) * std::pow(exps[i], L + 1.5) / dfact(2 * l - 1) /
                            dfact(2 * m - 1) / dfact(2 * n - 1) / std::pow(PI, 1.5));
    double prefactor = std::pow(PI, 1.5) * dfact(2 * l - 1) * dfact(2 * m - 1) * dfact(2 * n - 1) / std::pow(2.0, L);
    double N = 0.0;
    for (int i = 0; i < nprim; ++i)
        for (int j = 0; j < nprim; ++j)
            N += norm[i] * norm[j] * coefs[i] * coefs[j] / std::pow(exps[i] + exps[j], L + 1.5);
    N = 1 / std::sqrt(prefactor * N);
    for (int i = 0; i < nprim; ++i) coefs[i] *= N;
}

// Two-term McMurchie-Davidson recurrence, i raised first with j = 0, then j (pyx:988-1019).
// Every (i,j) sub-table is produced on the way, so one call serves all Cartesian components of a shell pair.
void hermite_table(int l1, int l2, double R, double a, double b, double *E)
{
    const int nt = l1 + l2 + 1;
    const double p = a + b, mu = a * b / p, half_inv_p = 1.0 / (2.0 * p);
    const double shift1 = -mu * R / a, shift2 = mu * R / b;
    auto at = [&](int i, int j, int t) -> double & { return E[(i * (l2 + 1) + j) * nt + t]; };
    for (int k = 0; k < (l1 + 1) * (l2 + 1) * nt; ++k) E[k] = 0.0;
    at(0, 0, 0) = std::exp(-mu * R * R);
    for (int i = 0; i <= l1; ++i)
        for (int j = 0; j <= l2; ++j) {
            if (i == 0 && j == 0) continue;
            const int pi = (j == 0) ? i - 1 : i, pj = (j == 0) ? 0 : j - 1;
            const double sh = (j == 0) ? shift1 : shift2;
            for (int t = 0; t <= i + j; ++t) {
                double up = (t + 1 < nt) ? at(pi, pj, t + 1) : 0.0;
                double v = sh * at(pi, pj, t) + (t + 1) * up;
                if (t > 0) v += half_inv_p * at(pi, pj, t - 1);
                at(i, j, t) = v;
            }
        }
}

// ---- Cartesian -> real spherical harmonics (closed form; reference tables kernel:554-623) ---------

static double fact(int n)
{
    double r = 1.0;
    for (int k = 2; k <= n; ++k) r *= k;
    return r;
}
static double binom(int n, int k)
{
    if (k < 0 || k > n) return 0.0;
    return fact(n) / (fact(k) * fact(n - k));
}

static std::complex<double> complex_coeff(int L, int m, int lx, int ly, int lz)
{
    const int am = std::abs(m);
    const int j2 = lx + ly - am;
    if (j2 < 0 || (j2 & 1)) return {0.0, 0.0};
    const int j = j2 / 2;
    double pref = std::sqrt(fact(2 * lx) * fact(2 * ly) * fact(2 * lz) * fact(L) * fact(L - am) /
                            (fact(2 * L) * fact(lx) * fact(ly) * fact(lz) * fact(L + am)));
    pref /= std::pow(2.0, L) * fact(L);
    double s1 = 0.0;
    for (int i = 0; i <= (L - am) / 2; ++i) {
        if (j > i) continue;
        s1 += binom(L, i) * binom(i, j) * ((i & 1) ? -1.0 : 1.0) * fact(2 * L - 2 * i) / fact(L - am - 2 * i);
    }
    std::complex<double> s2(0.0, 0.0);
    static const std::complex<double> ipow[4] = {{1, 0}, {0, 1}, {-1, 0}, {0, -1}};
    for (int k = 0; k <= j; ++k) {
        int q = lx - 2 * k;
        if (q < 0 || q > am) continue;
        int e = am - lx + 2 * k;
        s2 += binom(j, k) * binom(am, q) * ipow[((e % 4) + 4) % 4];
    }
    return pref * s1 * s2;
}

void sph_block(int L, std::vector<double> &U)
{
    const int nc = (L + 1) * (L + 2) / 2, ns = 2 * L + 1;
    U.assign((size_t)ns * nc, 0.0);
    if (L == 0) { U[0] = 1.0; return; }
    if (L == 1) { for (int k = 0; k < 3; ++k) U[k * 3 + k] = 1.0; return; }   // (x,y,z), kernel:556
    std::vector<int> order;
    if (L == 2) order = {-2, 1, -1, 2, 0};                                     // xy, xz, yz, x2-y2, z2 (kernel:562-568)
    else for (int m = -L; m <= L; ++m) order.push_back(m);
    for (int r = 0; r < ns; ++r) {
        const int m = order[r];
        int c = 0;
        for (int i = L; i >= 0; --i)
            for (int jj = L - i; jj >= 0; --jj, ++c) {
                std::complex<double> z = complex_coeff(L, m, i, jj, L - i - jj);
                double v = (m == 0) ? z.real() : (m > 0 ? std::sqrt(2.0) * z.real() : std::sqrt(2.0) * z.imag());
                U[(size_t)r * nc + c] = v;
            }
    }
}

// ---- Boys Taylor grid ------------------------------------------------------------------------------
// F_m(T0) for T0 = i/8, m < NORD, by the all-positive series at the top order and downward recursion,
// in long double, so the tabulated values are good to ~1e-19.
static void boys_table_compute(std::vector<double> &tab)
{
    tab.assign((size_t)TF_BOYS_NGRID * TF_BOYS_NORD, 0.0);
    const int top = TF_BOYS_NORD - 1;
    for (int i = 0; i < TF_BOYS_NGRID; ++i) {
        long double T = (long double)i * (long double)TF_BOYS_STEP;
        long double F[TF_BOYS_NORD];
        long double term = 1.0L / (2.0L * top + 1.0L), sum = term;
        for (int k = 1; k < 2000; ++k) {
            term *= 2.0L * T / (2.0L * top + 2.0L * k + 1.0L);
            sum += term;
            if (term < 1e-22L * sum) break;
        }
        long double e = expl(-T);
        F[top] = e * sum;
        for (int m = top; m > 0; --m) F[m - 1] = (2.0L * T * F[m] + e) / (2.0L * m - 1.0L);
        for (int m = 0; m < TF_BOYS_NORD; ++m) tab[(size_t)i * TF_BOYS_NORD + m] = (double)F[m];
    }
}

// the table is a constant of the library: computed once per process (~1 ms of long double arithmetic)
void boys_table(std::vector<double> &tab)
{
    static const std::vector<double> cached = [] { std::vector<double> t; boys_table_compute(t); return t; }();
    tab = cached;
}

// ---- basis ---------------------------------------------------------------------------------------

std::string build_basis(Basis &bs, int n, const double *origin, const int32_t *lmn, const int32_t *prim_off,
                        const double *exps, const double *coefs_raw)
{
    bs = Basis();
    if (n <= 0) return "n_ao_cart must be positive";
    bs.n_cart = n;
    bs.ao_origin.assign(origin, origin + 3 * (size_t)n);
    bs.ao_lmn.assign(lmn, lmn + 3 * (size_t)n);
    bs.ao_prim_off.assign(prim_off, prim_off + n + 1);
    const int ntot = prim_off[n];
    if (prim_off[0] != 0 || ntot <= 0) return "prim_off must start at 0 and be increasing";
    bs.ao_exp.assign(exps, exps + ntot);
    bs.ao_coef_raw.assign(coefs_raw, coefs_raw + ntot);
    bs.ao_coef = bs.ao_coef_raw;
    bs.ao_norm.assign(ntot, 0.0);
    bs.ao_shell.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        const int a = prim_off[i], k = prim_off[i + 1] - a;
        if (k <= 0) return "an AO has no primitives";
        for (int c = 0; c < 3; ++c)
            if (lmn[3 * i + c] < 0) return "negative angular momentum";
        if (lmn[3 * i] + lmn[3 * i + 1] + lmn[3 * i + 2] > TF_MAX_L) return "Only up to \"H\" type basis functions are implemented!";
        if (origin[3 * i] != 0.0 || origin[3 * i + 1] != 0.0)
            return "Molecule is incorrectly aligned! Unable to calculate molecular integrals.";
        normalize_ao(lmn[3 * i], lmn[3 * i + 1], lmn[3 * i + 2], k, &bs.ao_exp[a], &bs.ao_coef[a], &bs.ao_norm[a]);
    }
    // distinct centres -> atom index
    std::vector<double> centres;
    auto atom_of = [&](double z) {
        for (size_t k = 0; k < centres.size(); ++k)
            if (centres[k] == z) return (int)k;
        centres.push_back(z);
        return (int)centres.size() - 1;
    };
    // group consecutive AOs into shells
    int i = 0;
    while (i < n) {
        const int L = lmn[3 * i] + lmn[3 * i + 1] + lmn[3 * i + 2];
        const int nc = (L + 1) * (L + 2) / 2;
        const int a0 = prim_off[i], k0 = prim_off[i + 1] - a0;
        bool full = (i + nc <= n);
        if (full) {
            int c = 0;
            for (int x = L; x >= 0 && full; --x)
                for (int y = L - x; y >= 0 && full; --y, ++c) {
                    const int q = i + c, aq = prim_off[q];
                    if (lmn[3 * q] != x || lmn[3 * q + 1] != y || lmn[3 * q + 2] != L - x - y) full = false;
                    else if (prim_off[q + 1] - aq != k0 || origin[3 * q + 2] != origin[3 * i + 2]) full = false;
                    else if (std::memcmp(&exps[aq], &exps[a0], sizeof(double) * k0) ||
                             std::memcmp(&coefs_raw[aq], &coefs_raw[a0], sizeof(double) * k0)) full = false;
                }
        }
        Shell sh;
        sh.z = origin[3 * i + 2];
        sh.atom = atom_of(sh.z);
        sh.L = L;
        sh.nprim = k0;
        sh.prim_off = (int)bs.s_exp.size();
        sh.ncomp = full ? nc : 1;
        sh.comp_off = (int)bs.c_lx.size();
        sh.cart_off = i;
        sh.full = full;
        sh.nsph = full ? 2 * L + 1 : 1;
        sh.sph_off = 0;
        if (!full) bs.all_full = false;
        for (int p = 0; p < k0; ++p) {
            bs.s_exp.push_back(exps[a0 + p]);
            bs.s_w.push_back(bs.ao_norm[a0 + p] * bs.ao_coef[a0 + p]);
        }
        const double df0 = dfact(2 * lmn[3 * i] - 1) * dfact(2 * lmn[3 * i + 1] - 1) * dfact(2 * lmn[3 * i + 2] - 1);
        for (int c = 0; c < sh.ncomp; ++c) {
            const int q = i + c;
            bs.c_lx.push_back((int8_t)lmn[3 * q]);
            bs.c_ly.push_back((int8_t)lmn[3 * q + 1]);
            bs.c_lz.push_back((int8_t)lmn[3 * q + 2]);
            const double dfc = dfact(2 * lmn[3 * q] - 1) * dfact(2 * lmn[3 * q + 1] - 1) * dfact(2 * lmn[3 * q + 2] - 1);
            bs.c_scale.push_back(std::sqrt(df0 / dfc));
            bs.ao_shell[q] = (int)bs.shells.size();
        }
        bs.shells.push_back(sh);
        i += sh.ncomp;
    }
    if (centres.size() > 2) return "atoms and diatomics only (more than two centres found)";
    // spherical offsets + AO-level CSR of U
    int so = 0;
    for (auto &sh : bs.shells) { sh.sph_off = so; so += sh.nsph; }
    bs.n_sph = bs.all_full ? so : bs.n_cart;
    // shell pairs A >= B with their primitive-pair data
    const int ns = (int)bs.shells.size();
    std::vector<double> tmp;
    for (int A = 0; A < ns; ++A)
        for (int B = 0; B <= A; ++B) {
            const Shell &sa = bs.shells[A], &sb = bs.shells[B];
            Pair pr;
            pr.A = A; pr.B = B; pr.La = sa.L; pr.Lb = sb.L;
            pr.npp = sa.nprim * sb.nprim;
            pr.pp_off = (int)bs.pp_p.size();
            pr.nE = (sa.L + 1) * (sb.L + 1) * (sa.L + sb.L + 1);
            pr.e_off = (long long)bs.epool.size();
            const double AB = sa.z - sb.z;                     // pyx:1073
            tmp.resize(pr.nE);
            for (int a = 0; a < sa.nprim; ++a)
                for (int b = 0; b < sb.nprim; ++b) {
                    const double ea = bs.s_exp[sa.prim_off + a], eb = bs.s_exp[sb.prim_off + b], p = ea + eb;
                    bs.pp_p.push_back(p);
                    bs.pp_Pz.push_back((ea * sa.z + eb * sb.z) / p);   // pyx:1072
                    bs.pp_K.push_back(bs.s_w[sa.prim_off + a] * bs.s_w[sb.prim_off + b]);
                    bs.pp_AB.push_back(AB);
                    hermite_table(sa.L, sb.L, 0.0, ea, eb, tmp.data());   // x and y share this table
                    bs.epool.insert(bs.epool.end(), tmp.begin(), tmp.end());
                    hermite_table(sa.L, sb.L, AB, ea, eb, tmp.data());
                    bs.epool.insert(bs.epool.end(), tmp.begin(), tmp.end());
                }
            bs.pairs.push_back(pr);
        }
    return "";
}

void dense_sph_matrix(const Basis &bs, std::vector<double> &U)
{
    U.assign((size_t)bs.n_sph * bs.n_cart, 0.0);
    if (!bs.all_full) {
        for (int i = 0; i < bs.n_cart; ++i) U[(size_t)i * bs.n_cart + i] = 1.0;
        return;
    }
    std::vector<double> blk;
    for (const auto &sh : bs.shells) {
        sph_block(sh.L, blk);
        for (int r = 0; r < sh.nsph; ++r)
            for (int c = 0; c < sh.ncomp; ++c)
                U[(size_t)(sh.sph_off + r) * bs.n_cart + sh.cart_off + c] = blk[(size_t)r * sh.ncomp + c];
    }
}

}  // namespace tf
