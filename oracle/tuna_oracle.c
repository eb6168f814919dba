/*
 * tuna_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * A plain-C CPU restatement of the algorithm of the reference integral engine
 * (h-brough/TUNA v0.12.0, TUNA/tuna_integrals/tuna_integral.pyx, "pyx" below) for
 * z-axis diatomics.  It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function here
 * against the compiled, unmodified reference engine (oracle/_ref, built by
 * oracle/build_ref.sh from the sources where they lie) and against the golden vectors
 * that engine produced (tests/golden/).
 *
 * The one place this file cannot follow the reference literally is the Boys function:
 * pyx:1505 calls scipy.special.cython_special.hyp1f1 (SciPy, container version 1.15.3;
 * pyproject.toml:7 requires >= 1.15.0).  Here F_M(T) = 1F1(M+1/2; M+3/2; -T)/(2M+1) is
 * evaluated from its published series (Kummer-transformed, all terms positive), or for
 * large T from erf + upward recursion; the remaining orders use the reference's own
 * downward recursion (pyx:1565-1572).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.141592653589793238462643383279
#define ORC_PI32 5.5683279968317078452848179821188357 /* pi^1.5, pyx:14 */

typedef struct {
    const double *origin; /* 3 */
    int l, m, n;
    int nprim;
    const double *exps, *coefs, *norm;
} orc_bf;

/* pyx:245-271 */
static double double_fact(int n)
{
    double r = 1.0;
    if (n <= 0) return 1.0;
    while (n > 1) { r *= n; n -= 2; }
    return r;
}

/* pyx:174-210  Basis.normalize: primitive norms, then contracted renormalisation (coefs in place). */
void orc_normalize(int l, int m, int n, int nprim, const double *exps, double *coefs, double *norm)
{
    int L = l + m + n;
    for (int i = 0; i < nprim; ++i)
        norm[i] = sqrt(pow(2, 2 * L + 1.5) * pow(exps[i], L + 1.5) / double_fact(2 * l - 1) /
                       double_fact(2 * m - 1) / double_fact(2 * n - 1) / pow(ORC_PI, 1.5));
    double prefactor = pow(ORC_PI, 1.5) * double_fact(2 * l - 1) * double_fact(2 * m - 1) *
                       double_fact(2 * n - 1) / pow(2.0, L);
    double N = 0.0;
    for (int i = 0; i < nprim; ++i)
        for (int j = 0; j < nprim; ++j)
            N += norm[i] * norm[j] * coefs[i] * coefs[j] / pow(exps[i] + exps[j], L + 1.5);
    N = 1 / sqrt(prefactor * N);
    for (int i = 0; i < nprim; ++i) coefs[i] *= N;
}

/* pyx:1428-1481  recursive Hermite expansion coefficient E^{l1,l2}_t */
static double hermite_coeff(int l1, int l2, int t, double R, double a, double b)
{
    double p = a + b, u = a * b / p, pref = 1.0 / (2.0 * p), r;
    if (t < 0 || t > l1 + l2) return 0.0;
    if (l1 == 0 && l2 == 0 && t == 0) return exp(-u * R * R);
    if (l2 == 0) {
        r = pref * hermite_coeff(l1 - 1, l2, t - 1, R, a, b);
        r += -(u * R / a) * hermite_coeff(l1 - 1, l2, t, R, a, b);
        r += (t + 1) * hermite_coeff(l1 - 1, l2, t + 1, R, a, b);
    } else {
        r = pref * hermite_coeff(l1, l2 - 1, t - 1, R, a, b);
        r += (u * R / b) * hermite_coeff(l1, l2 - 1, t, R, a, b);
        r += (t + 1) * hermite_coeff(l1, l2 - 1, t + 1, R, a, b);
    }
    return r;
}

/* Boys function of top order M (stands in for pyx:1490-1505, see header). */
static double boys_top(int M, double T)
{
    if (T < 45.0 + M) {
        /* F_M(T) = exp(-T) * sum_k (2T)^k / ((2M+1)(2M+3)...(2M+2k+1)) */
        double term = 1.0 / (2.0 * M + 1.0), sum = term;
        for (int k = 1; k < 400; ++k) {
            term *= 2.0 * T / (2.0 * M + 2.0 * k + 1.0);
            sum += term;
            if (term < 1e-18 * sum) break;
        }
        return exp(-T) * sum;
    }
    /* large T: F_0 = sqrt(pi/T)/2 * erf(sqrt T); F_{m+1} = ((2m+1) F_m - exp(-T)) / (2T) */
    double F = 0.5 * sqrt(ORC_PI / T) * erf(sqrt(T)), e = exp(-T);
    for (int m = 0; m < M; ++m) F = ((2 * m + 1) * F - e) / (2.0 * T);
    return F;
}

double orc_boys(int m, double T) { return (T == 0.0) ? 1.0 / (2.0 * m + 1.0) : boys_top(m, T); }

/* pyx:1540-1572 */
static void fill_boys_table(int M, double T, double *tab)
{
    if (T == 0.0) {
        for (int m = 0; m <= M; ++m) tab[m] = 1.0 / (2.0 * m + 1.0);
        return;
    }
    tab[M] = boys_top(M, T);
    double e = exp(-T), two_T = 2.0 * T;
    for (int m = M; m > 0; --m) tab[m - 1] = (two_T * tab[m] + e) / (2.0 * m - 1.0);
}

/* pyx:1582-1602 */
static void fill_pow_table(int M, double scale, double *tab)
{
    double f = -2.0 * scale;
    tab[0] = 1.0;
    for (int n = 1; n <= M; ++n) tab[n] = tab[n - 1] * f;
}

/* pyx:1612-1651  z-only Hermite Coulomb table R[v][n] */
static void fill_Rz(int Vmax, int Nmax, double PCz, const double *boys, const double *pw, double *R)
{
    int stride = Nmax + 1;
    for (int n = 0; n <= Nmax; ++n) R[n] = pw[n] * boys[n];
    for (int v = 1; v <= Vmax; ++v)
        for (int n = Nmax - v; n >= 0; --n) {
            R[v * stride + n] = PCz * R[(v - 1) * stride + n + 1];
            if (v > 1) R[v * stride + n] += (v - 1) * R[(v - 2) * stride + n + 1];
        }
}

/* pyx:446-615  overlap, kinetic, dipole, diagonal quadrupole for one contracted pair */
static void local_integrals(const orc_bf *b1, const orc_bf *b2, const double *origin, double *out /*8*/)
{
    int l1 = b1->l, m1 = b1->m, n1 = b1->n, l2 = b2->l, m2 = b2->m, n2 = b2->n;
    double dx = b1->origin[0] - b2->origin[0], dy = b1->origin[1] - b2->origin[1],
           dz = b1->origin[2] - b2->origin[2];
    double s = 0, t = 0, d_x = 0, d_y = 0, d_z = 0, qx = 0, qy = 0, qz = 0;
    for (int i = 0; i < b1->nprim; ++i) {
        double a = b1->exps[i], pa = b1->norm[i] * b1->coefs[i];
        for (int j = 0; j < b2->nprim; ++j) {
            double b = b2->exps[j], pb = b2->norm[j] * b2->coefs[j];
            double p = a + b;
            double pref = pa * pb * ORC_PI32 / (p * sqrt(p));
            double Sx = hermite_coeff(l1, l2, 0, dx, a, b), Sy = hermite_coeff(m1, m2, 0, dy, a, b),
                   Sz = hermite_coeff(n1, n2, 0, dz, a, b);
            double Ex1 = hermite_coeff(l1, l2, 1, dx, a, b), Ey1 = hermite_coeff(m1, m2, 1, dy, a, b),
                   Ez1 = hermite_coeff(n1, n2, 1, dz, a, b);
            double Ex2 = hermite_coeff(l1, l2, 2, dx, a, b), Ey2 = hermite_coeff(m1, m2, 2, dy, a, b),
                   Ez2 = hermite_coeff(n1, n2, 2, dz, a, b);
            double Ax = (2 * l2 + 1) * b, Ay = (2 * m2 + 1) * b, Az = (2 * n2 + 1) * b;
            double Bx = -0.5 * l2 * (l2 - 1), By = -0.5 * m2 * (m2 - 1), Bz = -0.5 * n2 * (n2 - 1);
            double Tx = Ax * Sx - 2.0 * b * b * hermite_coeff(l1, l2 + 2, 0, dx, a, b) +
                        Bx * hermite_coeff(l1, l2 - 2, 0, dx, a, b);
            double Ty = Ay * Sy - 2.0 * b * b * hermite_coeff(m1, m2 + 2, 0, dy, a, b) +
                        By * hermite_coeff(m1, m2 - 2, 0, dy, a, b);
            double Tz = Az * Sz - 2.0 * b * b * hermite_coeff(n1, n2 + 2, 0, dz, a, b) +
                        Bz * hermite_coeff(n1, n2 - 2, 0, dz, a, b);
            double Px = (a * b1->origin[0] + b * b2->origin[0]) / p - origin[0];
            double Py = (a * b1->origin[1] + b * b2->origin[1]) / p - origin[1];
            double Pz = (a * b1->origin[2] + b * b2->origin[2]) / p - origin[2];
            double Dx = Ex1 + Px * Sx, Dy = Ey1 + Py * Sy, Dz = Ez1 + Pz * Sz;
            double Qx = 2.0 * Ex2 + 2.0 * Px * Ex1 + (Px * Px + 1.0 / (2.0 * p)) * Sx;
            double Qy = 2.0 * Ey2 + 2.0 * Py * Ey1 + (Py * Py + 1.0 / (2.0 * p)) * Sy;
            double Qz = 2.0 * Ez2 + 2.0 * Pz * Ez1 + (Pz * Pz + 1.0 / (2.0 * p)) * Sz;
            s += pref * Sx * Sy * Sz;
            t += pref * (Tx * Sy * Sz + Sx * Ty * Sz + Sx * Sy * Tz);
            d_x += pref * Dx * Sy * Sz;
            d_y += pref * Sx * Dy * Sz;
            d_z += pref * Sx * Sy * Dz;
            qx += pref * Qx * Sy * Sz;
            qy += pref * Sx * Qy * Sz;
            qz += pref * Sx * Sy * Qz;
        }
    }
    out[0] = s; out[1] = t; out[2] = d_x; out[3] = d_y; out[4] = d_z; out[5] = qx; out[6] = qy; out[7] = qz;
}

/* pyx:779-891  nuclear attraction <1| 1/r_C |2> for a nucleus on the z axis */
static double nuclear_integral(const orc_bf *b1, const orc_bf *b2, const double *C)
{
    int l1 = b1->l, m1 = b1->m, n1 = b1->n, l2 = b2->l, m2 = b2->m, n2 = b2->n;
    double z1 = b1->origin[2], z2 = b2->origin[2], zc = C[2], R12 = z1 - z2;
    int Vmax = n1 + n2, Nmax = l1 + l2 + m1 + m2 + n1 + n2, stride = Nmax + 1;
    double boys[64], pw[64], Rz[1024];
    double integral = 0.0;
    for (int i = 0; i < b1->nprim; ++i) {
        double a = b1->exps[i], pa = b1->norm[i] * b1->coefs[i];
        for (int j = 0; j < b2->nprim; ++j) {
            double b = b2->exps[j], pb = b2->norm[j] * b2->coefs[j];
            double p = a + b, PCz = (a * z1 + b * z2) / p - zc;
            fill_boys_table(Nmax, p * PCz * PCz, boys);
            fill_pow_table(Nmax, p, pw);
            fill_Rz(Vmax, Nmax, PCz, boys, pw, Rz);
            double prim = 0.0;
            for (int t = 0; t <= l1 + l2; t += 2) {
                double Ex = hermite_coeff(l1, l2, t, 0.0, a, b) * double_fact(t - 1);
                for (int u = 0; u <= m1 + m2; u += 2) {
                    double Ey = hermite_coeff(m1, m2, u, 0.0, a, b) * double_fact(u - 1);
                    for (int v = 0; v <= n1 + n2; ++v) {
                        double Ez = hermite_coeff(n1, n2, v, R12, a, b);
                        prim += Ex * Ey * Ez * Rz[v * stride + (t + u) / 2];
                    }
                }
            }
            integral += pa * pb * prim * 2.0 * ORC_PI / p;
        }
    }
    return integral;
}

static void unpack_bfs(int n, const double *origin, const int *lmn, const int *prim_off, const double *exps,
                       const double *coefs, const double *norm, orc_bf *bfs)
{
    for (int i = 0; i < n; ++i) {
        bfs[i].origin = origin + 3 * i;
        bfs[i].l = lmn[3 * i]; bfs[i].m = lmn[3 * i + 1]; bfs[i].n = lmn[3 * i + 2];
        bfs[i].nprim = prim_off[i + 1] - prim_off[i];
        bfs[i].exps = exps + prim_off[i];
        bfs[i].coefs = coefs + prim_off[i];
        bfs[i].norm = norm + prim_off[i];
    }
}

/* pyx:282-435  S, T, V, D[3], Q[3] (Cartesian AOs).  coefs are the NORMALISED coefficients. */
void orc_one_electron(int n, const double *origin, const int *lmn, const int *prim_off, const double *exps,
                      const double *coefs, const double *norm, int n_atoms, const double *atom_xyz,
                      const double *atom_charge, const double *dip_origin, double *S, double *T, double *V,
                      double *D, double *Q, int num_threads)
{
    orc_bf *bfs = (orc_bf *)malloc(sizeof(orc_bf) * n);
    unpack_bfs(n, origin, lmn, prim_off, exps, coefs, norm, bfs);
    (void)num_threads;
#pragma omp parallel for schedule(guided) num_threads(num_threads)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double o[8];
            local_integrals(&bfs[i], &bfs[j], dip_origin, o);
            double v = 0.0;
            for (int a = 0; a < n_atoms; ++a)
                v = v - nuclear_integral(&bfs[i], &bfs[j], atom_xyz + 3 * a) * atom_charge[a];
            size_t ij = (size_t)i * n + j, ji = (size_t)j * n + i, nn = (size_t)n * n;
            S[ij] = S[ji] = o[0];
            T[ij] = T[ji] = o[1];
            V[ij] = V[ji] = v;
            for (int c = 0; c < 3; ++c) {
                D[c * nn + ij] = D[c * nn + ji] = o[2 + c];
                Q[c * nn + ij] = Q[c * nn + ji] = o[5 + c];
            }
        }
    free(bfs);
}

/* pyx:626-768  overlap between two different basis sets */
void orc_cross_overlap(int n1, const double *origin1, const int *lmn1, const int *off1, const double *exps1,
                       const double *coefs1, const double *norm1, int n2, const double *origin2,
                       const int *lmn2, const int *off2, const double *exps2, const double *coefs2,
                       const double *norm2, double *S)
{
    orc_bf *A = (orc_bf *)malloc(sizeof(orc_bf) * n1), *B = (orc_bf *)malloc(sizeof(orc_bf) * n2);
    unpack_bfs(n1, origin1, lmn1, off1, exps1, coefs1, norm1, A);
    unpack_bfs(n2, origin2, lmn2, off2, exps2, coefs2, norm2, B);
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            double dx = A[i].origin[0] - B[j].origin[0], dy = A[i].origin[1] - B[j].origin[1],
                   dz = A[i].origin[2] - B[j].origin[2], s = 0.0;
            for (int k = 0; k < A[i].nprim; ++k)
                for (int l = 0; l < B[j].nprim; ++l) {
                    double a = A[i].exps[k], b = B[j].exps[l], p = a + b;
                    double pref = A[i].norm[k] * A[i].coefs[k] * B[j].norm[l] * B[j].coefs[l] * ORC_PI32 /
                                  (p * sqrt(p));
                    s = s + pref * hermite_coeff(A[i].l, B[j].l, 0, dx, a, b) *
                                hermite_coeff(A[i].m, B[j].m, 0, dy, a, b) *
                                hermite_coeff(A[i].n, B[j].n, 0, dz, a, b);
                }
            S[(size_t)i * n2 + j] = s;
        }
    free(A); free(B);
}

/* ---- two-electron integrals ------------------------------------------------------------- */

typedef struct { /* pyx:35-44 */
    double coefficient, exponent_sum, product_centre_z, centre_distance_z;
    double hx[20], hy[20], hz[20];
} prim_pair;

typedef struct { /* pyx:49-67 */
    int i, j, lx_sum, ly_sum, lz_sum, t_start, u_start, npp;
    prim_pair *pp;
} ao_pair;

/* pyx:961-1036  iterative two-term recurrence; i raised first (j = 0), then j */
static void hermite_table_iter(int l1, int l2, double R, double a, double b, double *tab, int use_parity)
{
    int n_l2 = l2 + 1, nterms = l1 + l2 + 1, stride = nterms + 1;
    int total = (l1 + 1) * (l2 + 1) * stride;
    double p = a + b, mu = a * b / p, pref = 1.0 / (2.0 * p);
    double shift1 = -mu * R / a, shift2 = mu * R / b;
    double E[8400];
    for (int k = 0; k < total; ++k) E[k] = 0.0;
    E[0] = exp(-mu * R * R);
    for (int i = 0; i <= l1; ++i)
        for (int j = 0; j <= l2; ++j) {
            if (i == 0 && j == 0) continue;
            int base = (i * n_l2 + j) * stride;
            int prev = (j == 0) ? ((i - 1) * n_l2 + j) * stride : (i * n_l2 + j - 1) * stride;
            double sh = (j == 0) ? shift1 : shift2;
            for (int t = 0; t <= i + j; ++t) {
                E[base + t] = sh * E[prev + t] + (t + 1) * E[prev + t + 1];
                if (t > 0) E[base + t] += pref * E[prev + t - 1];
            }
        }
    int base = (l1 * n_l2 + l2) * stride;
    if (use_parity) {
        for (int t = 0; t < nterms; ++t) tab[t] = 0.0;
        for (int t = (l1 + l2) & 1; t < nterms; t += 2) tab[t] = E[base + t];
    } else
        for (int t = 0; t < nterms; ++t) tab[t] = E[base + t];
}

/* pyx:1050-1128 */
static void build_ao_pair(ao_pair *P, int i, int j, const orc_bf *b1, const orc_bf *b2)
{
    P->i = i; P->j = j;
    P->lx_sum = b1->l + b2->l; P->ly_sum = b1->m + b2->m; P->lz_sum = b1->n + b2->n;
    P->t_start = P->lx_sum & 1; P->u_start = P->ly_sum & 1;
    P->npp = b1->nprim * b2->nprim;
    P->pp = (prim_pair *)malloc(sizeof(prim_pair) * (size_t)P->npp);
    int k = 0;
    for (int a = 0; a < b1->nprim; ++a)
        for (int b = 0; b < b2->nprim; ++b, ++k) {
            prim_pair *q = &P->pp[k];
            double ea = b1->exps[a], eb = b2->exps[b], p = ea + eb;
            q->coefficient = b1->norm[a] * b2->norm[b] * b1->coefs[a] * b2->coefs[b];
            q->exponent_sum = p;
            q->product_centre_z = (ea * b1->origin[2] + eb * b2->origin[2]) / p;
            q->centre_distance_z = b1->origin[2] - b2->origin[2];
            hermite_table_iter(b1->l, b2->l, 0.0, ea, eb, q->hx, 1);
            hermite_table_iter(b1->m, b2->m, 0.0, ea, eb, q->hy, 1);
            hermite_table_iter(b1->n, b2->n, q->centre_distance_z, ea, eb, q->hz, 0);
        }
}

/* pyx:1142-1221 */
static double primitive_quartet(const ao_pair *A, const prim_pair *a, const ao_pair *B, const prim_pair *b)
{
    double p = a->exponent_sum, q = b->exponent_sum, pq = p + q, alpha = p * q / pq;
    double PQz = a->product_centre_z - b->product_centre_z;
    int Vmax = A->lz_sum + B->lz_sum;
    int Nmax = A->lx_sum + A->ly_sum + A->lz_sum + B->lx_sum + B->ly_sum + B->lz_sum, stride = Nmax + 1;
    double boys[64], pw[64], Rz[1024], integral = 0.0;
    fill_boys_table(Nmax, alpha * PQz * PQz, boys);
    fill_pow_table(Nmax, alpha, pw);
    fill_Rz(Vmax, Nmax, PQz, boys, pw, Rz);
    for (int t = A->t_start; t <= A->lx_sum; t += 2)
        for (int tau = B->t_start; tau <= B->lx_sum; tau += 2) {
            double xf = a->hx[t] * b->hx[tau] * double_fact(t + tau - 1);
            for (int u = A->u_start; u <= A->ly_sum; u += 2)
                for (int nu = B->u_start; nu <= B->ly_sum; nu += 2) {
                    double xyf = xf * a->hy[u] * b->hy[nu] * double_fact(u + nu - 1);
                    int nxy = ((t + tau) >> 1) + ((u + nu) >> 1);
                    for (int v = 0; v <= A->lz_sum; ++v) {
                        double c12 = a->hz[v];
                        if (c12 == 0.0) continue;
                        for (int phi = 0; phi <= B->lz_sum; ++phi) {
                            double c34 = b->hz[phi];
                            if (c34 == 0.0) continue;
                            double sign = ((tau + nu + phi) & 1) ? -1.0 : 1.0;
                            integral += xyf * c12 * c34 * sign * Rz[(v + phi) * stride + nxy];
                        }
                    }
                }
        }
    double pref = 34.986836655249725 / (p * q * sqrt(pq)); /* 2 pi^(5/2), pyx:1219 */
    return a->coefficient * b->coefficient * pref * integral;
}

/* pyx:1267-1355  full dense (ij|kl) tensor, Cartesian AOs, all 8 images written */
void orc_eri(int n, const double *origin, const int *lmn, const int *prim_off, const double *exps,
             const double *coefs, const double *norm, double *ERI, int num_threads)
{
    orc_bf *bfs = (orc_bf *)malloc(sizeof(orc_bf) * n);
    unpack_bfs(n, origin, lmn, prim_off, exps, coefs, norm, bfs);
    long npair = (long)n * (n + 1) / 2;
    ao_pair *pairs = (ao_pair *)malloc(sizeof(ao_pair) * (size_t)npair);
    long k = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) build_ao_pair(&pairs[k++], i, j, &bfs[i], &bfs[j]);
    size_t N = (size_t)n;
    (void)num_threads;
#pragma omp parallel for schedule(dynamic) num_threads(num_threads)
    for (long p12 = 0; p12 < npair; ++p12) {
        const ao_pair *A = &pairs[p12];
        size_t i = A->i, j = A->j;
        for (long p34 = 0; p34 <= p12; ++p34) {
            const ao_pair *B = &pairs[p34];
            size_t kk = B->i, l = B->j;
            double val = 0.0;
            if (!(((A->lx_sum + B->lx_sum) & 1) || ((A->ly_sum + B->ly_sum) & 1)))
                for (int a = 0; a < A->npp; ++a)
                    for (int b = 0; b < B->npp; ++b) val += primitive_quartet(A, &A->pp[a], B, &B->pp[b]);
            ERI[((i * N + j) * N + kk) * N + l] = val;
            ERI[((kk * N + l) * N + i) * N + j] = val;
            ERI[((j * N + i) * N + l) * N + kk] = val;
            ERI[((l * N + kk) * N + j) * N + i] = val;
            ERI[((j * N + i) * N + kk) * N + l] = val;
            ERI[((l * N + kk) * N + i) * N + j] = val;
            ERI[((i * N + j) * N + l) * N + kk] = val;
            ERI[((kk * N + l) * N + j) * N + i] = val;
        }
    }
    for (long q = 0; q < npair; ++q) free(pairs[q].pp);
    free(pairs); free(bfs);
}

/* single contracted integral (pyx:1376-1414) */
double orc_eri_element(const double *origin, const int *lmn, const int *prim_off, const double *exps,
                       const double *coefs, const double *norm)
{
    orc_bf b[4];
    unpack_bfs(4, origin, lmn, prim_off, exps, coefs, norm, b);
    ao_pair A, B;
    build_ao_pair(&A, 0, 0, &b[0], &b[1]);
    build_ao_pair(&B, 0, 0, &b[2], &b[3]);
    double val = 0.0;
    if (!(((A.lx_sum + B.lx_sum) & 1) || ((A.ly_sum + B.ly_sum) & 1)))
        for (int a = 0; a < A.npp; ++a)
            for (int c = 0; c < B.npp; ++c) val += primitive_quartet(&A, &A.pp[a], &B, &B.pp[c]);
    free(A.pp); free(B.pp);
    return val;
}
