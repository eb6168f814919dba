// tf_dft.hip.h -- Kohn-Sham exchange-correlation on the GPU (BASELINE config 4: CO B3LYP/def2-TZVP).
// AOs and their gradients are evaluated once on the molecular grid; per SCF iteration the density, its gradient, the
// functional derivatives and the V_XC matrix are two GEMMs (rocBLAS) around element-wise kernels:
//     B = Phi P            rho_g = <B_g, Phi_g>     grad_a = 2 <B_g, dPhi_a,g>
//     D = w (vrho Phi + 4 vsigma sum_a grad_a dPhi_a)          V_XC = sym(Phi^T D)
// Reference: construct_basis_functions_on_grid tuna_dft.py:516-584, construct_basis_function_gradients_on_grid :586-666,
// construct_density_on_grid :677-703, calculate_density_gradient :714-744, calculate_V_X/V_C :788-888,
// calculate_restricted_exchange_correlation_matrix tuna_scf.py:600-654, functionals tuna_xc.py (Slater :199-213, B88 :385-438,
// B3 :1462-1494, VWN :1512-1630 + :1802-1860, LYP :2200-2260, 3P :5843-5881), floors tuna_util.py:95-99.
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <string>
#include <vector>

#include "../../include/tunafock.h"
#include "tf_internal.h"

namespace tfdft {

enum { X_NONE = 0, X_SLATER = 1, X_B88 = 2, X_B3 = 3 };
enum { C_NONE = 0, C_VWN5 = 1, C_VWN3 = 2, C_LYP = 3, C_3P_VWN5 = 4, C_3P_VWN3 = 5 };

struct Grid {
    long long G = 0;
    int N = 0, xid = 0, cid = 0;
    bool gga = false;
    double dfx = 0, dfc = 0, x_alpha = 2.0 / 3.0;
    double *w = nullptr, *phi = nullptr, *dphi = nullptr;        // [G], [G][N], [3][G][N]
    double *B = nullptr, *D = nullptr;                           // [G][N]
    double *rho = nullptr, *grad = nullptr, *vrho = nullptr, *vsig = nullptr, *ex = nullptr, *ec = nullptr;   // [G], [3][G], ...
    double *V = nullptr, *part = nullptr;                        // [VSPLIT + 1][N][N] split-K partials of V, reduction partials
};

inline void release(Grid &g)
{
    for (double *p : {g.w, g.phi, g.dphi, g.B, g.D, g.rho, g.grad, g.vrho, g.vsig, g.ex, g.ec, g.V, g.part})
        if (p) (void)hipFree(p);
    g = Grid();
}

struct DAOs { const double *z; const int *lmn, *prim_off; const double *exps, *w; };

// phi[g][i], dphi[a][g][i] for OUTPUT AO i = sum over its CSR row of Cartesian AOs (identity row for Cartesian output)
__global__ void ao_on_grid_kernel(DAOs A, const double *__restrict__ xyz, long long G, int N, const int *__restrict__ ptr,
                                  const int *__restrict__ idx, const double *__restrict__ val, double *__restrict__ phi,
                                  double *__restrict__ dphi, int with_grad)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= G * N) return;
    const long long g = e / N;
    const int i = (int)(e - g * N);
    const double X = xyz[g], Y = xyz[G + g], Z = xyz[2 * G + g];
    double f = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
    for (int q = ptr[i]; q < ptr[i + 1]; ++q) {
        const int c = idx[q];
        const int l = A.lmn[3 * c], m = A.lmn[3 * c + 1], n = A.lmn[3 * c + 2];
        const double zr = Z - A.z[c];
        const double r2 = X * X + Y * Y + zr * zr;
        double px = 1.0, py = 1.0, pz = 1.0;
        for (int k = 0; k < l; ++k) px *= X;
        for (int k = 0; k < m; ++k) py *= Y;
        for (int k = 0; k < n; ++k) pz *= zr;
        double pxm = 1.0, pym = 1.0, pzm = 1.0;                 // x^(l-1), ...
        for (int k = 0; k + 1 < l; ++k) pxm *= X;
        for (int k = 0; k + 1 < m; ++k) pym *= Y;
        for (int k = 0; k + 1 < n; ++k) pzm *= zr;
        const double poly = px * py * pz;
        const double dpx = l > 0 ? l * pxm * py * pz : 0.0, dpy = m > 0 ? m * px * pym * pz : 0.0, dpz = n > 0 ? n * px * py * pzm : 0.0;
        double s = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
        for (int p = A.prim_off[c]; p < A.prim_off[c + 1]; ++p) {
            const double a = A.exps[p];
            const double ew = A.w[p] * exp(-a * r2);
            s += ew;
            if (with_grad) {
                sx += ew * (dpx - 2.0 * a * X * poly);
                sy += ew * (dpy - 2.0 * a * Y * poly);
                sz += ew * (dpz - 2.0 * a * zr * poly);
            }
        }
        f += val[q] * s * poly;
        fx += val[q] * sx; fy += val[q] * sy; fz += val[q] * sz;
    }
    phi[e] = f;
    if (with_grad) { dphi[e] = fx; dphi[G * N + e] = fy; dphi[2 * G * N + e] = fz; }
}

// ---- functionals (restricted, closed shell) -------------------------------------------------------------------------

struct XcOut { double dfdn, dfds, e; };

__device__ inline XcOut slater_x(double n, double x_alpha)                     // tuna_xc.py:199-213
{
    XcOut o;
    o.dfdn = -(3.0 / 2.0 * x_alpha) * cbrt(3.0 / 3.141592653589793 * n);
    o.e = 3.0 / 4.0 * o.dfdn;
    o.dfds = 0.0;
    return o;
}

__device__ inline XcOut b88_x(double n, double sigma, double x_alpha)          // tuna_xc.py:385-438
{
    const double beta = 0.0042;
    const double C = 2.0 / cbrt(4.0);
    const double eLDA = slater_x(n / 2.0, x_alpha).e;
    const double c3 = cbrt(n / 2.0);
    const double x = sqrt(sigma / 4.0) / (c3 * c3 * c3 * c3);
    const double x2 = x * x;
    const double A = asinh(x);
    const double Dd = 1.0 + 6.0 * beta * x * A;
    const double D2 = Dd * Dd;
    const double dDdx = 6.0 * beta * (A + x / sqrt(1.0 + x2));
    XcOut o;
    o.e = C * eLDA - beta * c3 * x2 / Dd;
    o.dfdn = (o.e + C * eLDA / 3.0 + beta * c3 * (7.0 * x2 * Dd - 4.0 * x2 * x * dDdx) / (3.0 * D2));
    o.dfds = -beta * n * c3 * (x2 * Dd - (1.0 / 2.0) * x2 * x * dDdx) / (sigma * D2);
    return o;
}

__device__ inline XcOut vwn_c(double n, double x_0, double b, double c, double A)   // tuna_xc.py:1802-1860
{
    const double Q = sqrt(4.0 * c - b * b);
    const double X_0 = x_0 * x_0 + b * x_0 + c;
    const double c_1 = -b * x_0 / X_0;
    const double c_2 = 2.0 * b * (c - x_0 * x_0) / (Q * X_0);
    const double r_s = cbrt(3.0 / (4.0 * 3.141592653589793) * (1.0 / n));
    const double x = sqrt(r_s);
    const double xm = x - x_0;
    const double Xx = r_s + b * x + c;
    const double log1 = log(r_s / Xx);
    const double log2 = log(xm * xm / Xx);
    const double at = atan(Q / (2.0 * x + b));
    const double combo = (2.0 / x + 2.0 * c_1 / xm - (2.0 * x + b) * (1.0 + c_1) / Xx - (1.0 / 2.0) * c_2 * Q / Xx);
    XcOut o;
    o.e = A * (log1 + c_1 * log2 + c_2 * at);
    const double dedr = (A / 2.0) * combo / x;
    o.dfdn = o.e - r_s / 3.0 * dedr;
    o.dfds = 0.0;
    return o;
}

__device__ inline XcOut lyp_c(double n, double sigma)                           // tuna_xc.py:2200-2260
{
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349;
    const double PI = 3.141592653589793;
    const double inv_n = 1.0 / n;
    const double c3 = cbrt(n);
    const double ic3 = 1.0 / c3;
    const double X = 1.0 + d * ic3;
    const double k = cbrt(3.0 * PI * PI);
    const double c3_2 = c3 * c3, c3_4 = c3_2 * c3_2, c3_8 = c3_4 * c3_4;
    const double C2 = 6.0 / 10.0 * k * k * c3_8;
    const double ic3_2 = ic3 * ic3, ic3_4 = ic3_2 * ic3_2, ic3_8 = ic3_4 * ic3_4;
    const double w = ic3_8 * ic3_2 * ic3 * exp(-c * ic3) / X;              // inv_cbrt^11
    const double delta = ic3 * (c + d / X);
    const double mabw = -a * b * w * n;
    const double wpw = -(1.0 / 3.0) * ic3_4 * (11.0 * c3 - c - d / X);
    const double dprime = (1.0 / 3.0) * (d * d * ic3_4 * ic3 / (X * X) - delta * inv_n);
    XcOut o;
    o.dfds = mabw * n * (-7.0 * delta - 3.0) / 72.0;
    double dfdn = -a / X + mabw * sigma * (-1.0 / 12.0 - 7.0 * delta / 36.0 + n * (-7.0 * dprime / 72.0 + wpw * (-1.0 / 24.0 - 7.0 * delta / 72.0)));
    dfdn += n * (-a * d / (3.0 * X * X * c3_4) - 7.0 * C2 * a * b * w / 3.0 - (1.0 / 2.0) * C2 * a * b * n * wpw * w);
    o.dfdn = dfdn;
    o.e = (1.0 / 2.0) * C2 * mabw - mabw * sigma * (7.0 * delta + 3.0) / 72.0 - a / X;
    return o;
}

// rho (raw) and 2 grad rho from B = Phi P: rho(g) = sum_i B[g][i] Phi[g][i], grad_a = 2 sum_i B[g][i] dPhi_a[g][i].  Sixteen lanes per grid
// point, each a sixteenth of the AOs (coalesced 128-byte reads of the point's rows; one thread per point walked its own row of every
// array with a stride of N doubles between neighbouring threads: 0.30 ms per call for CO / def2-TZVP, 3.6 ms of a 27 ms single point).
__global__ __launch_bounds__(256) void xc_density_kernel(long long G, int N, const double *__restrict__ phi, const double *__restrict__ dphi,
                                                         const double *__restrict__ B, int gga, double *__restrict__ rho, double *__restrict__ grad)
{
    const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
    const long long g = (long long)blockIdx.x * 16 + grp;
    double n = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
    if (g < G) {
        const double *b = B + g * N, *f = phi + g * N;
        for (int i = l; i < N; i += 16) n += b[i] * f[i];
        if (gga) {
            const double *fx = dphi + g * N, *fy = dphi + G * N + g * N, *fz = dphi + 2 * G * N + g * N;
            for (int i = l; i < N; i += 16) { const double bi = b[i]; gx += bi * fx[i]; gy += bi * fy[i]; gz += bi * fz[i]; }
        }
    }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) {
        n += __shfl_xor(n, d, 16); gx += __shfl_xor(gx, d, 16); gy += __shfl_xor(gy, d, 16); gz += __shfl_xor(gz, d, 16);
    }
    if (l == 0 && g < G) {
        rho[g] = n;
        if (gga) { grad[g] = 2.0 * gx; grad[G + g] = 2.0 * gy; grad[2 * G + g] = 2.0 * gz; }
    }
}

// rho (floored), sigma (floored) of a point from xc_density_kernel's sums; then functional derivatives and energy densities
__global__ void xc_point_kernel(long long G, int gga, int xid, int cid, double dfx, double dfc, double x_alpha,
                                double *__restrict__ rho, const double *__restrict__ grad, double *__restrict__ vrho, double *__restrict__ vsig,
                                double *__restrict__ ex, double *__restrict__ ec)
{
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double n = rho[g], gx = 0.0, gy = 0.0, gz = 0.0;
    if (gga) { gx = grad[g]; gy = grad[G + g]; gz = grad[2 * G + g]; }
    n = fmax(n, 1e-23);                                                     // xc.clean, density_floor
    const double sigma = fmax(gx * gx + gy * gy + gz * gz, 1e-46);          // sigma_floor
    XcOut X{0, 0, 0}, C{0, 0, 0};
    if (xid == X_SLATER) X = slater_x(n, x_alpha);
    else if (xid == X_B88) X = b88_x(n, sigma, x_alpha);
    else if (xid == X_B3) {                                                 // 0.9 B88 + 0.1 Slater, tuna_xc.py:1462-1494
        const XcOut s = slater_x(n, x_alpha), bb = b88_x(n, sigma, x_alpha);
        X.dfdn = 0.9 * bb.dfdn + 0.1 * s.dfdn; X.dfds = 0.9 * bb.dfds; X.e = 0.9 * bb.e + 0.1 * s.e;
    }
    if (cid == C_VWN5) C = vwn_c(n, -0.10498, 3.72744, 12.9352, 0.0310907);
    else if (cid == C_VWN3) C = vwn_c(n, -0.409286, 13.0720, 42.7198, 0.0310907);
    else if (cid == C_LYP) C = lyp_c(n, sigma);
    else if (cid == C_3P_VWN5 || cid == C_3P_VWN3) {                        // 0.81 LYP + 0.19 VWN, tuna_xc.py:5843-5881
        const XcOut l = (cid == C_3P_VWN5) ? vwn_c(n, -0.10498, 3.72744, 12.9352, 0.0310907) : vwn_c(n, -0.409286, 13.0720, 42.7198, 0.0310907);
        const XcOut y = lyp_c(n, sigma);
        C.dfdn = 0.81 * y.dfdn + 0.19 * l.dfdn; C.dfds = 0.81 * y.dfds; C.e = 0.81 * y.e + 0.19 * l.e;
    }
    rho[g] = n;
    vrho[g] = dfx * X.dfdn + dfc * C.dfdn;
    vsig[g] = dfx * X.dfds + dfc * C.dfds;
    ex[g] = X.e * n;
    ec[g] = C.e * n;
}

// D[g][n] = w (vrho phi + 4 vsig sum_a grad_a dphi_a)
__global__ void xc_dmat_kernel(long long G, int N, const double *__restrict__ w, const double *__restrict__ phi,
                               const double *__restrict__ dphi, const double *__restrict__ grad, const double *__restrict__ vrho,
                               const double *__restrict__ vsig, int gga, double *__restrict__ D)
{
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= G * N) return;
    const long long g = e / N;
    double v = vrho[g] * phi[e];
    if (gga) v += 4.0 * vsig[g] * (grad[g] * dphi[e] + grad[G + g] * dphi[G * N + e] + grad[2 * G + g] * dphi[2 * G * N + e]);
    D[e] = w[g] * v;
}

// partial sums of w*rho, w*ex, w*ec
__global__ void xc_reduce_kernel(long long G, const double *__restrict__ w, const double *__restrict__ rho, const double *__restrict__ ex,
                                 const double *__restrict__ ec, double *__restrict__ part)
{
    __shared__ double s0[256], s1[256], s2[256];
    double a = 0, b = 0, c = 0;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += (long long)gridDim.x * blockDim.x) {
        a += w[g] * rho[g]; b += w[g] * ex[g]; c += w[g] * ec[g];
    }
    s0[threadIdx.x] = a; s1[threadIdx.x] = b; s2[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { s0[threadIdx.x] += s0[threadIdx.x + s]; s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = s0[0]; part[3 * blockIdx.x + 1] = s1[0]; part[3 * blockIdx.x + 2] = s2[0]; }
}

// tmp = sum_s in[s] over the nparts split-K partials (fixed order; coalesced), then out = sym(tmp)
__global__ void sum_parts_kernel(const double *__restrict__ in, int nparts, double *__restrict__ tmp, int nn)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nn) return;
    double a = 0.0;
    for (int s = 0; s < nparts; ++s) a += in[(size_t)s * nn + e];
    tmp[e] = a;
}
__global__ void sym_kernel(const double *__restrict__ tmp, double *__restrict__ out, int n)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    out[e] = (1.0 / 2.0) * (tmp[e] + tmp[(size_t)j * n + i]);
}

// V = Phi^T D contracts over ~10^5 grid points into an N x N matrix: as ONE GEMM rocBLAS runs it in a single workgroup (4.7 ms for CO /
// def2-TZVP); split over the grid points into VSPLIT batch entries it fills the GPU, and the partials are added in fixed order.
const int VSPLIT = 128;            // (512 partials of 200 grid points each cost more to add up -- 0.18 ms -- than the GEMM they split)

#define TFD_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { msg = std::string(#call) + " failed: " + hipGetErrorString(_e); return (_e == hipErrorOutOfMemory ? TF_ENOMEM : TF_ENODEVICE); } } while (0)
#define TFD_BLAS(call) do { rocblas_status _s = (call); if (_s != rocblas_status_success) { msg = std::string(#call) + " failed (rocBLAS status " + std::to_string((int)_s) + ")"; return TF_ELINALG; } } while (0)

const int NPART = 512;

// V_XC (device, [N][N]) and the integrals {n_elec, E_X*dfx, E_C*dfc} for the device density dP
inline int vxc(rocblas_handle blas, Grid &g, const double *dP, double *dVxc, double out3[3], std::string &msg)
{
    const int N = g.N;
    const long long G = g.G;
    const double one = 1.0, zero = 0.0;
    // B (G x N, row-major) = Phi (G x N) * P (N x N)
    TFD_BLAS(rocblas_dgemm(blas, rocblas_operation_none, rocblas_operation_none, N, (rocblas_int)G, N, &one, dP, N, g.phi, N, &zero, g.B, N));
    hipLaunchKernelGGL(xc_density_kernel, dim3((unsigned)((G + 15) / 16)), dim3(256), 0, 0, G, N, g.phi, g.dphi, g.B, g.gga ? 1 : 0, g.rho, g.grad);
    hipLaunchKernelGGL(xc_point_kernel, dim3((unsigned)((G + 127) / 128)), dim3(128), 0, 0, G, g.gga ? 1 : 0, g.xid, g.cid,
                       g.dfx, g.dfc, g.x_alpha, g.rho, g.grad, g.vrho, g.vsig, g.ex, g.ec);
    hipLaunchKernelGGL(xc_dmat_kernel, dim3((unsigned)((G * N + 255) / 256)), dim3(256), 0, 0, G, N, g.w, g.phi, g.dphi, g.grad, g.vrho, g.vsig,
                       g.gga ? 1 : 0, g.D);
    // V (N x N, row-major) = Phi^T (N x G) * D (G x N), split over the grid points
    {
        const long long Kc = G / VSPLIT, rem = G - Kc * VSPLIT;
        const size_t nn = (size_t)N * N;
        int nparts = 0;
        if (Kc > 0) {
            TFD_BLAS(rocblas_dgemm_strided_batched(blas, rocblas_operation_none, rocblas_operation_transpose, N, N, (rocblas_int)Kc, &one, g.D, N,
                                                   (rocblas_stride)(Kc * N), g.phi, N, (rocblas_stride)(Kc * N), &zero, g.V, N, (rocblas_stride)nn,
                                                   VSPLIT));
            nparts = VSPLIT;
        }
        if (rem > 0) {
            TFD_BLAS(rocblas_dgemm(blas, rocblas_operation_none, rocblas_operation_transpose, N, N, (rocblas_int)rem, &one, g.D + Kc * VSPLIT * N, N,
                                   g.phi + Kc * VSPLIT * N, N, &zero, g.V + (size_t)nparts * nn, N));
            ++nparts;
        }
        double *tmp = g.V + (size_t)(VSPLIT + 1) * nn;                  // (the slot behind the partials)
        hipLaunchKernelGGL(sum_parts_kernel, dim3((N * N + 255) / 256), dim3(256), 0, 0, g.V, nparts, tmp, N * N);
        hipLaunchKernelGGL(sym_kernel, dim3((N * N + 255) / 256), dim3(256), 0, 0, tmp, dVxc, N);
    }
    hipLaunchKernelGGL(xc_reduce_kernel, dim3(NPART), dim3(256), 0, 0, G, g.w, g.rho, g.ex, g.ec, g.part);
    std::vector<double> h(3 * NPART);
    TFD_HIP(hipMemcpy(h.data(), g.part, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    out3[0] = out3[1] = out3[2] = 0.0;
    for (int b = 0; b < NPART; ++b) { out3[0] += h[3 * b]; out3[1] += h[3 * b + 1]; out3[2] += h[3 * b + 2]; }
    out3[1] *= g.dfx; out3[2] *= g.dfc;
    return TF_OK;
}

}  // namespace tfdft
