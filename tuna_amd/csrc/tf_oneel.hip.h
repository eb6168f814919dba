// tf_oneel.hip.h -- one-electron companions on the GPU: S, T, V, dipole, diagonal quadrupole and the
// cross-basis overlap.  O(N^2 K^2) work, one thread per AO pair; nothing here is performance critical,
// it exists so that a whole SCF can run without any CPU integral code.
// Reference: calculate_one_electron_integrals pyx:282-435, calculate_contracted_local_integrals pyx:446-615,
// calculate_contracted_nuclear_integral pyx:779-891, calculate_cross_basis_overlap_matrix pyx:626-768.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "tf_internal.h"

namespace tfone {

struct DAO {
    const double *z;        // centre (z axis)
    const int *lmn;         // 3n
    const int *prim_off;    // n+1
    const double *exps, *w; // w = norm * normalised coefficient
};

#define TF1_LMAX1 (TF_MAX_L + 1)
#define TF1_LMAX2 (TF_MAX_L + 3)
#define TF1_NT (2 * TF_MAX_L + 3)
#define TF1_TAB (TF1_LMAX1 * TF1_LMAX2 * TF1_NT)

// E[i][j][t] for i <= l1, j <= l2: same recurrence as the reference's recursive hermite_coeff (pyx:1428-1481),
// i raised first (j = 0), then j; terms added in the reference's order.
__device__ inline void hermite_fill(int l1, int l2, double R, double a, double b, double *E)
{
    const double p = a + b, mu = a * b / p, pref = 1.0 / (2.0 * p);
    const double sh1 = -(mu * R / a), sh2 = (mu * R / b);
    for (int k = 0; k < (l1 + 1) * TF1_LMAX2 * TF1_NT; ++k) E[k] = 0.0;
    E[0] = exp(-mu * R * R);
    for (int i = 0; i <= l1; ++i)
        for (int j = 0; j <= l2; ++j) {
            if (i == 0 && j == 0) continue;
            const int pi = (j == 0) ? i - 1 : i, pj = (j == 0) ? 0 : j - 1;
            const double sh = (j == 0) ? sh1 : sh2;
            const double *prev = E + (pi * TF1_LMAX2 + pj) * TF1_NT;
            double *cur = E + (i * TF1_LMAX2 + j) * TF1_NT;
            for (int t = 0; t <= i + j; ++t) {
                double r = (t > 0) ? pref * prev[t - 1] : 0.0;
                r += sh * prev[t];
                r += (t + 1) * ((t + 1 < TF1_NT) ? prev[t + 1] : 0.0);
                cur[t] = r;
            }
        }
}

__device__ inline double Eget(const double *E, int i, int j, int t)
{
    if (j < 0 || t < 0 || t > i + j) return 0.0;
    return E[(i * TF1_LMAX2 + j) * TF1_NT + t];
}

__device__ inline double dfact_odd(int n)  // n!! (n <= 0 -> 1)
{
    double r = 1.0;
    while (n > 1) { r *= n; n -= 2; }
    return r;
}

// Boys table F_0..F_M(T) in a private array: Taylor grid at the top order + downward recursion (pyx:1540-1572).
__device__ inline void boys_fill(int M, double T, const double *__restrict__ tab, double *F)
{
    if (T == 0.0) {
        for (int m = 0; m <= M; ++m) F[m] = 1.0 / (2.0 * m + 1.0);
        return;
    }
    if (T < TF_BOYS_TMAX) {
        const int i = (int)(T * (1.0 / TF_BOYS_STEP) + 0.5);
        const double d = (double)i * TF_BOYS_STEP - T;
        const double *row = tab + (size_t)i * TF_BOYS_NORD + M;
        double f = row[8];
        for (int k = 8; k >= 1; --k) f = row[k - 1] + f * d / (double)k;
        const double e = exp(-T), two_T = 2.0 * T;
        F[M] = f;
        for (int m = M; m > 0; --m) F[m - 1] = (two_T * F[m] + e) / (2.0 * m - 1.0);
    } else {
        const double e = exp(-T), inv2T = 1.0 / (2.0 * T);
        F[0] = 0.5 * sqrt(3.141592653589793238462643383279 / T);
        for (int m = 0; m < M; ++m) F[m + 1] = ((2.0 * m + 1.0) * F[m] - e) * inv2T;
    }
}

// WAVE_PER_PAIR: one wave per AO pair, the lanes share the primitive pairs (a deeply contracted pair -- 13 x 13 primitives of two Ar s
// shells -- kept one thread busy for milliseconds) and are summed in a fixed butterfly at the end.  Otherwise one LANE per AO pair
// (uncontracted or barely contracted basis sets: 118 000 AO pairs of one primitive pair each at N = 400 -- a wave per pair left 63 of
// 64 lanes idle, 8 ms; now 64 pairs per wave).
template <bool WAVE_PER_PAIR>
__global__ void oneel_kernel(DAO A, int n, int n_atoms, double zc0, double zc1, double q0, double q1, double oz,
                             const double *__restrict__ boys, double *__restrict__ S, double *__restrict__ T,
                             double *__restrict__ V, double *__restrict__ D, double *__restrict__ Q)
{
    const long long pidx = WAVE_PER_PAIR ? (long long)blockIdx.x : (long long)blockIdx.x * 64 + threadIdx.x;
    const int lane = WAVE_PER_PAIR ? (int)threadIdx.x : 0;
    const long long npair = (long long)n * (n + 1) / 2;
    if (pidx >= npair) return;
    int i = (int)((sqrt(8.0 * (double)pidx + 1.0) - 1.0) * 0.5);
    while ((long long)i * (i + 1) / 2 > pidx) --i;
    while ((long long)(i + 1) * (i + 2) / 2 <= pidx) ++i;
    const int j = (int)(pidx - (long long)i * (i + 1) / 2);

    const int l1 = A.lmn[3 * i], m1 = A.lmn[3 * i + 1], n1 = A.lmn[3 * i + 2];
    const int l2 = A.lmn[3 * j], m2 = A.lmn[3 * j + 1], n2 = A.lmn[3 * j + 2];
    const double z1 = A.z[i], z2 = A.z[j], dz = z1 - z2;
    const int L1 = l1 + m1 + n1, L2 = l2 + m2 + n2;
    double Exy[TF1_TAB], Ez[TF1_TAB];
    double F[2 * TF_MAX_L + 2], Rz[(2 * TF_MAX_L + 1) * (2 * TF_MAX_L + 1)];
    double s = 0, t = 0, dx = 0, dy = 0, dzz = 0, qx = 0, qy = 0, qz = 0, v0 = 0, v1 = 0;
    const double PI = 3.141592653589793238462643383279, PI32 = 5.5683279968317078452848179821188357;
    const int Vmax = n1 + n2, Nmax = L1 + L2, stride = Nmax + 1;
    const int pa0 = A.prim_off[i], npa = A.prim_off[i + 1] - pa0, pb0 = A.prim_off[j], npb = A.prim_off[j + 1] - pb0;
    for (int pq = lane; pq < npa * npb; pq += (WAVE_PER_PAIR ? 64 : 1)) {
        {
            const int a = pa0 + pq / npb, b = pb0 + pq % npb;
            const double ea = A.exps[a], wa = A.w[a];
            const double eb = A.exps[b], wb = A.w[b];
            const double p = ea + eb;
            const double pref = wa * wb * PI32 / (p * sqrt(p));
            hermite_fill(L1, L2 + 2, 0.0, ea, eb, Exy);
            hermite_fill(L1, L2 + 2, dz, ea, eb, Ez);
            const double Sx = Eget(Exy, l1, l2, 0), Sy = Eget(Exy, m1, m2, 0), Sz = Eget(Ez, n1, n2, 0);
            const double Ex1 = Eget(Exy, l1, l2, 1), Ey1 = Eget(Exy, m1, m2, 1), Ez1 = Eget(Ez, n1, n2, 1);
            const double Ex2 = Eget(Exy, l1, l2, 2), Ey2 = Eget(Exy, m1, m2, 2), Ez2 = Eget(Ez, n1, n2, 2);
            const double Ax = (2 * l2 + 1) * eb, Ay = (2 * m2 + 1) * eb, Az = (2 * n2 + 1) * eb;
            const double Bx = -0.5 * l2 * (l2 - 1), By = -0.5 * m2 * (m2 - 1), Bz = -0.5 * n2 * (n2 - 1);
            const double Tx = Ax * Sx - 2.0 * eb * eb * Eget(Exy, l1, l2 + 2, 0) + Bx * Eget(Exy, l1, l2 - 2, 0);
            const double Ty = Ay * Sy - 2.0 * eb * eb * Eget(Exy, m1, m2 + 2, 0) + By * Eget(Exy, m1, m2 - 2, 0);
            const double Tz = Az * Sz - 2.0 * eb * eb * Eget(Ez, n1, n2 + 2, 0) + Bz * Eget(Ez, n1, n2 - 2, 0);
            const double Px = 0.0, Py = 0.0;                                  // atoms and origin on the z axis
            const double Pz = (ea * z1 + eb * z2) / p - oz;
            const double Dx = Ex1 + Px * Sx, Dy = Ey1 + Py * Sy, Dz = Ez1 + Pz * Sz;
            const double Qx = 2.0 * Ex2 + 2.0 * Px * Ex1 + (Px * Px + 1.0 / (2.0 * p)) * Sx;
            const double Qy = 2.0 * Ey2 + 2.0 * Py * Ey1 + (Py * Py + 1.0 / (2.0 * p)) * Sy;
            const double Qz = 2.0 * Ez2 + 2.0 * Pz * Ez1 + (Pz * Pz + 1.0 / (2.0 * p)) * Sz;
            s += pref * Sx * Sy * Sz;
            t += pref * (Tx * Sy * Sz + Sx * Ty * Sz + Sx * Sy * Tz);
            dx += pref * Dx * Sy * Sz; dy += pref * Sx * Dy * Sz; dzz += pref * Sx * Sy * Dz;
            qx += pref * Qx * Sy * Sz; qy += pref * Sx * Qy * Sz; qz += pref * Sx * Sy * Qz;
            // nuclear attraction, one nucleus at a time (pyx:857-885)
            for (int at = 0; at < n_atoms; ++at) {
                const double PC = (ea * z1 + eb * z2) / p - (at == 0 ? zc0 : zc1);
                boys_fill(Nmax, p * PC * PC, boys, F);
                double pw = 1.0;
                for (int nn = 0; nn <= Nmax; ++nn) { Rz[nn] = pw * F[nn]; pw *= -2.0 * p; }
                for (int v = 1; v <= Vmax; ++v)
                    for (int nn = Nmax - v; nn >= 0; --nn) {
                        double r = PC * Rz[(v - 1) * stride + nn + 1];
                        if (v > 1) r += (v - 1) * Rz[(v - 2) * stride + nn + 1];
                        Rz[v * stride + nn] = r;
                    }
                double prim = 0.0;
                for (int tt = 0; tt <= l1 + l2; tt += 2) {
                    const double ex = Eget(Exy, l1, l2, tt) * dfact_odd(tt - 1);
                    for (int u = 0; u <= m1 + m2; u += 2) {
                        const double ey = Eget(Exy, m1, m2, u) * dfact_odd(u - 1);
                        for (int v = 0; v <= n1 + n2; ++v) prim += ex * ey * Eget(Ez, n1, n2, v) * Rz[v * stride + (tt + u) / 2];
                    }
                }
                const double contrib = wa * wb * prim * 2.0 * PI / p;
                if (at == 0) v0 += contrib; else v1 += contrib;
            }
        }
    }
    if (WAVE_PER_PAIR)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off); t += __shfl_xor(t, off); dx += __shfl_xor(dx, off); dy += __shfl_xor(dy, off); dzz += __shfl_xor(dzz, off);
        qx += __shfl_xor(qx, off); qy += __shfl_xor(qy, off); qz += __shfl_xor(qz, off); v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off);
    }
    if (lane != 0) return;
    double v = 0.0;                         // pyx:391-395: v = v - integral * charge, atom by atom
    v = v - v0 * q0;
    if (n_atoms > 1) v = v - v1 * q1;
    const size_t ij = (size_t)i * n + j, ji = (size_t)j * n + i, nn2 = (size_t)n * n;
    S[ij] = S[ji] = s; T[ij] = T[ji] = t; V[ij] = V[ji] = v;
    D[ij] = D[ji] = dx; D[nn2 + ij] = D[nn2 + ji] = dy; D[2 * nn2 + ij] = D[2 * nn2 + ji] = dzz;
    Q[ij] = Q[ji] = qx; Q[nn2 + ij] = Q[nn2 + ji] = qy; Q[2 * nn2 + ij] = Q[2 * nn2 + ji] = qz;
}

__global__ void cross_overlap_kernel(DAO A, int n1, DAO B, int n2, double *__restrict__ S)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)n1 * n2) return;
    const int i = (int)(e / n2), j = (int)(e - (long long)i * n2);
    const int l1 = A.lmn[3 * i], m1 = A.lmn[3 * i + 1], nn1 = A.lmn[3 * i + 2];
    const int l2 = B.lmn[3 * j], m2 = B.lmn[3 * j + 1], nn2 = B.lmn[3 * j + 2];
    const double dz = A.z[i] - B.z[j];
    double Exy[TF1_TAB], Ez[TF1_TAB];
    const double PI32 = 5.5683279968317078452848179821188357;
    double s = 0.0;
    for (int a = A.prim_off[i]; a < A.prim_off[i + 1]; ++a)
        for (int b = B.prim_off[j]; b < B.prim_off[j + 1]; ++b) {
            const double ea = A.exps[a], eb = B.exps[b], p = ea + eb;
            const double pref = A.w[a] * B.w[b] * PI32 / (p * sqrt(p));
            hermite_fill(l1 + m1 + nn1, l2 + m2 + nn2, 0.0, ea, eb, Exy);
            hermite_fill(l1 + m1 + nn1, l2 + m2 + nn2, dz, ea, eb, Ez);
            s = s + pref * Eget(Exy, l1, l2, 0) * Eget(Exy, m1, m2, 0) * Eget(Ez, nn1, nn2, 0);
        }
    S[e] = s;
}

// out[i][j] = sum_ab U[i,a] U[j,b] M[a][b]   (kernel:495-502), U given as AO-level CSR
__global__ void sph_matrix_kernel(const double *__restrict__ M, int Nc, int Ns, const int *__restrict__ ptr,
                                  const int *__restrict__ idx, const double *__restrict__ val, double *__restrict__ out)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Ns * Ns) return;
    const int i = e / Ns, j = e - i * Ns;
    double s = 0.0;
    for (int qa = ptr[i]; qa < ptr[i + 1]; ++qa) {
        double t = 0.0;
        for (int qb = ptr[j]; qb < ptr[j + 1]; ++qb) t += val[qb] * M[(size_t)idx[qa] * Nc + idx[qb]];
        s += val[qa] * t;
    }
    out[e] = s;
}

// A block of device memory kept by the context for these short-lived buffers: a dozen hipMalloc / hipFree pairs per call cost more
// (~0.2 ms each) than the kernels of a small molecule.  Bump allocation; what does not fit is allocated separately and the block
// grows on the next call.
struct Arena {
    char *base = nullptr;
    size_t cap = 0, used = 0, wanted = 0;
    void release() { if (base) (void)hipFree(base); base = nullptr; cap = used = wanted = 0; }
};

struct DevBuf {
    std::vector<void *> ptrs;
    Arena *arena = nullptr;
    explicit DevBuf(Arena *a = nullptr) : arena(a)
    {
        if (!arena) return;
        if (arena->wanted > arena->cap) {
            if (arena->base) (void)hipFree(arena->base);
            arena->base = nullptr; arena->cap = 0;
            const size_t want = arena->wanted + arena->wanted / 4 + 4096;
            if (hipMalloc((void **)&arena->base, want) == hipSuccess) arena->cap = want;
        }
        arena->used = 0; arena->wanted = 0;
    }
    ~DevBuf() { for (void *p : ptrs) (void)hipFree(p); if (arena) arena->used = 0; }
    void *raw(size_t bytes, std::string &err)
    {
        bytes = (std::max<size_t>(1, bytes) + 255) & ~(size_t)255;
        if (arena) {
            arena->wanted += bytes;
            if (arena->used + bytes <= arena->cap) { void *p = arena->base + arena->used; arena->used += bytes; return p; }
        }
        void *d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) { err = "hipMalloc failed (one-electron)"; return nullptr; }
        ptrs.push_back(d);
        return d;
    }
    template <class T> T *put(const std::vector<T> &h, std::string &err)
    {
        T *d = static_cast<T *>(raw(h.size() * sizeof(T), err));
        if (d && !h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) { err = "hipMemcpy failed (one-electron)"; return nullptr; }
        return d;
    }
    template <class T> T *alloc(size_t n, std::string &err) { return static_cast<T *>(raw(n * sizeof(T), err)); }
};

inline DAO upload_aos(const tf::Basis &bs, DevBuf &buf, std::string &err)
{
    std::vector<double> z(bs.n_cart), w(bs.ao_exp.size());
    for (int i = 0; i < bs.n_cart; ++i) z[i] = bs.ao_origin[3 * i + 2];
    for (size_t k = 0; k < w.size(); ++k) w[k] = bs.ao_norm[k] * bs.ao_coef[k];
    std::vector<int> lmn(bs.ao_lmn.begin(), bs.ao_lmn.end()), off(bs.ao_prim_off.begin(), bs.ao_prim_off.end());
    DAO A;
    A.z = buf.put(z, err); A.lmn = buf.put(lmn, err); A.prim_off = buf.put(off, err);
    A.exps = buf.put(bs.ao_exp, err); A.w = buf.put(w, err);
    return A;
}

inline void sph_csr(const tf::Basis &bs, std::vector<int> &ptr, std::vector<int> &idx, std::vector<double> &val)
{
    ptr.assign(1, 0); idx.clear(); val.clear();
    std::vector<double> blk;
    for (const auto &sh : bs.shells) {
        tf::sph_block(sh.L, blk);
        for (int r = 0; r < sh.nsph; ++r) {
            for (int c = 0; c < sh.ncomp; ++c)
                if (blk[(size_t)r * sh.ncomp + c] != 0.0) { idx.push_back(sh.cart_off + c); val.push_back(blk[(size_t)r * sh.ncomp + c]); }
            ptr.push_back((int)idx.size());
        }
    }
}

inline std::string one_electron(const tf::Basis &bs, int n_atoms, const double *xyz, const double *charge, const double *origin,
                                int spherical, double *S, double *T, double *V, double *D, double *Q, const double *d_boys_cached = nullptr,
                                Arena *arena = nullptr)
{
    std::string err;
    DevBuf buf(arena);
    const int n = bs.n_cart;
    const size_t nn = (size_t)n * n;
    DAO A = upload_aos(bs, buf, err);
    const double *d_boys = d_boys_cached;                       // the context's copy (tf_set_basis) when there is one
    if (!d_boys) {
        std::vector<double> boys;
        tf::boys_table(boys);
        d_boys = buf.put(boys, err);
    }
    double *d_all = buf.alloc<double>(9 * nn, err);
    if (!err.empty()) return err;
    double *dS = d_all, *dT = d_all + nn, *dV = d_all + 2 * nn, *dD = d_all + 3 * nn, *dQ = d_all + 6 * nn;
    const long long npair = (long long)n * (n + 1) / 2;
    int max_nprim = 1;                                          // deepest contraction of an AO
    for (int i = 0; i < n; ++i) max_nprim = std::max(max_nprim, (int)(bs.ao_prim_off[i + 1] - bs.ao_prim_off[i]));
    if (max_nprim <= 2)                                         // (at most four primitive pairs per AO pair: a lane per pair)
        hipLaunchKernelGGL(oneel_kernel<false>, dim3((unsigned)((npair + 63) / 64)), dim3(64), 0, 0, A, n, n_atoms, xyz[2],
                           n_atoms > 1 ? xyz[5] : 0.0, charge[0], n_atoms > 1 ? charge[1] : 0.0, origin[2], d_boys, dS, dT, dV, dD, dQ);
    else
        hipLaunchKernelGGL(oneel_kernel<true>, dim3((unsigned)npair), dim3(64), 0, 0, A, n, n_atoms, xyz[2],
                           n_atoms > 1 ? xyz[5] : 0.0, charge[0], n_atoms > 1 ? charge[1] : 0.0, origin[2], d_boys, dS, dT, dV, dD, dQ);
    double *src = d_all;
    int m = n;
    if (spherical) {
        std::vector<int> ptr, idx; std::vector<double> val;
        sph_csr(bs, ptr, idx, val);
        int *dp = buf.put(ptr, err), *di = buf.put(idx, err);
        double *dv = buf.put(val, err);
        m = bs.n_sph;
        double *d_out = buf.alloc<double>(9 * (size_t)m * m, err);
        if (!err.empty()) return err;
        for (int k = 0; k < 9; ++k)
            hipLaunchKernelGGL(sph_matrix_kernel, dim3((m * m + 255) / 256), dim3(256), 0, 0, d_all + k * nn, n, m, dp, di, dv,
                               d_out + (size_t)k * m * m);
        src = d_out;
    }
    const size_t mm = (size_t)m * m;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(S, src, mm * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(T, src + mm, mm * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(V, src + 2 * mm, mm * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && D) e = hipMemcpy(D, src + 3 * mm, 3 * mm * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && Q) e = hipMemcpy(Q, src + 6 * mm, 3 * mm * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return std::string("one-electron integrals failed on the device: ") + hipGetErrorString(e);
    return "";
}

inline std::string cross_overlap(const tf::Basis &b1, const tf::Basis &b2, double *S, Arena *arena = nullptr)
{
    std::string err;
    DevBuf buf(arena);
    DAO A = upload_aos(b1, buf, err), B = upload_aos(b2, buf, err);
    const size_t tot = (size_t)b1.n_cart * b2.n_cart;
    double *dS = buf.alloc<double>(tot, err);
    if (!err.empty()) return err;
    hipLaunchKernelGGL(cross_overlap_kernel, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, 0, A, b1.n_cart, B, b2.n_cart, dS);
    hipError_t e = hipMemcpy(S, dS, tot * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return std::string("cross overlap failed on the device: ") + hipGetErrorString(e);
    return "";
}

}  // namespace tfone
