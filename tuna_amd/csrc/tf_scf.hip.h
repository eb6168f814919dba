// tf_scf.hip.h -- the restricted SCF cycle, resident on the GPU: J/K from the stored tensor (tf_kernels),
// rocBLAS for the O(N^3) products, rocSOLVER dsyevd for the symmetric eigenproblem; only scalars and the
// tiny DIIS system touch the host.
// Reference: run_restricted_SCF_cycle scf:1072-1154, run_self_consistent_field_cycle scf:1292-1435,
// calculate_DIIS_error scf:879-949, apply_DIIS scf:960-1061, apply_damping scf:763-868,
// diagonalise_Fock_matrix scf:222-250, construct_density_matrix scf:183-211,
// calculate_restricted_electronic_energy scf:344-404, calculate_orthogonalisation_matrix kernel:756-816.
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/tunafock.h"
#include "tf_jacobi.hip.h"
#include "tf_refine.hip.h"
#include <algorithm>
#include <cstdio>
#include <vector>

namespace tfscf {

// The stream the SCF code of the calling host thread works on: the legacy default stream, except inside a lockstep batch
// (tf_scf_rhf_batch), where every cycle runs on a host thread and a non-blocking stream of its own so that the O(N^3) steps of different
// cycles overlap on the device.
inline thread_local hipStream_t t_stream = nullptr;
#define TFS_ST (tfscf::t_stream)
inline hipError_t tfs_memcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    if (!t_stream) return hipMemcpy(dst, src, bytes, kind);
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, t_stream);
    return e != hipSuccess ? e : hipStreamSynchronize(t_stream);
}
inline hipError_t tfs_sync() { return t_stream ? hipStreamSynchronize(t_stream) : hipDeviceSynchronize(); }

struct Workspace {
    // Sharded tensors (world > 1): every rank runs the cycle redundantly on identical data, and a last-bit difference must not let one
    // rank leave the loop (or take another branch) while the others wait in the next all-reduce.  `agree` sums a few per-iteration
    // decision values over the ranks (tf_device.hip: through the registered all-reduce) and fails on EVERY rank when they differ.
    std::function<int(const double *vals, int n, std::string &msg)> agree;
    rocblas_handle blas = nullptr;
    int n = 0;
    double *pool = nullptr;      // all N x N device matrices live in one allocation
    size_t pool_doubles = 0;
    double *d_scal = nullptr;    // small device scalar array
    double *d_part = nullptr;    // [256][8] block partials of the two-stage reductions
    rocblas_int *d_info = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *jac_scratch = nullptr;   // eigenvector scratch of the Jacobi solver when it does not fit LDS
    size_t jac_cap = 0;
    double *jac_prev = nullptr;      // eigenvectors of the previous solve (warm start inside one SCF run)
    size_t jac_prev_cap = 0;
    int jac_prev_n = 0;              // 0 = no valid warm start
    bool warm_ok = false;            // set by run_rhf for the duration of one SCF cycle
    // eigenvector refinement (n > 64): rows of ref_X are the current orthonormal approximate eigenvectors, ref_lam their values
    double *ref_buf = nullptr;       // [7][n][n] + n doubles: X, Xnew, G, Y, S, E, scaled X; lambda
    size_t ref_cap = 0;
    int ref_n = 0;                   // 0 = no valid start
    long long ref_solves = 0, ref_steps = 0, ref_fallbacks = 0;
    double *h_pin = nullptr;         // pinned host buffer + event: the per-step read-back of the refinement overlaps the next GEMMs
    size_t h_pin_cap = 0;
    hipEvent_t ev_pin = nullptr;
    double *ref_alt_buf = nullptr;   // the same for the second spin of an unrestricted cycle (swapped in around its solves)
    size_t ref_alt_cap = 0;
    int ref_alt_n = 0;
    std::vector<hipEvent_t> tev;     // timing events, read after the cycle (no synchronisation inside it)
    // Symmetry blocks (eigh_blocked): class of every basis function (set_symmetry; empty = none known), the functions of each class
    std::vector<int> sym_cls;
    int sym_n = 0, sym_nb = 0, sym_mmax = 0;   // matrix size the tables were built for, classes present, largest class
    int sym_m[4] = {0, 0, 0, 0};
    int *sym_idx = nullptr;          // device [4][mmax]: original index of member t of block b (-1: padding); then cls[n]
    double *sym_buf = nullptr;       // device: [nb][mmax][mmax] blocks, [nb][mmax] values, [nb][mmax] work, 2 doubles of the cross-class test
    int *sym_src = nullptr;          // device [n]: block * mmax + member of the eigenvalue of global rank r
    rocblas_int *sym_info = nullptr; // device [4]
    // blocked refinement (ref_refine_blocks): every block <= 64 vectors -- the in-LDS refinement kernel on all blocks in one launch
    double *blk_X = nullptr, *blk_alt_X = nullptr;   // device [nb][mmax][mmax] compact: the blocks' current vectors (second slot: the other spin)
    int *blk_ints = nullptr;         // device: noccs[4] | status[4][2] | combined[2] | .. | noccs[4] of the other slot at 16
    int blk_ref_n = 0, blk_alt_ref_n = 0;            // n when blk_X / noccs are valid, else 0
    int blk_nocc_off = 0, blk_alt_nocc_off = 16;     // where this slot's noccs[4] live in blk_ints
    long long blk_solves = 0, blk_fallbacks = 0;
    bool sym_last_blocked = false;   // the last eigh() went block by block: sym_src names the block of every eigenvector it returned
    int ref_vcls_n = 0, ref_alt_vcls_n = 0;   // n when the refinement's vectors carry block labels (tail of ref_buf), else 0
    double *sym_prev = nullptr;      // device [nb][mmax][mmax]: block eigenvectors of the previous solve (warm start of the batched Jacobi)
    int sym_prev_n = 0;              // 0 = none
    long long sym_solves = 0, sym_declined = 0;
};

inline void release(Workspace &w)
{
    if (w.pool) (void)hipFree(w.pool);
    if (w.d_scal) (void)hipFree(w.d_scal);
    if (w.d_part) (void)hipFree(w.d_part);
    if (w.d_info) (void)hipFree(w.d_info);
    if (w.blas) (void)rocblas_destroy_handle(w.blas);
    if (w.ev0) (void)hipEventDestroy(w.ev0);
    if (w.ev1) (void)hipEventDestroy(w.ev1);
    if (w.jac_scratch) (void)hipFree(w.jac_scratch);
    if (w.jac_prev) (void)hipFree(w.jac_prev);
    if (w.ref_buf) (void)hipFree(w.ref_buf);
    if (w.ref_alt_buf) (void)hipFree(w.ref_alt_buf);
    if (w.h_pin) (void)hipHostFree(w.h_pin);
    if (w.ev_pin) (void)hipEventDestroy(w.ev_pin);
    for (hipEvent_t e : w.tev) (void)hipEventDestroy(e);
    if (w.sym_idx) (void)hipFree(w.sym_idx);
    if (w.sym_buf) (void)hipFree(w.sym_buf);
    if (w.sym_src) (void)hipFree(w.sym_src);
    if (w.sym_info) (void)hipFree(w.sym_info);
    if (w.sym_prev) (void)hipFree(w.sym_prev);
    if (w.blk_X) (void)hipFree(w.blk_X);
    if (w.blk_alt_X) (void)hipFree(w.blk_alt_X);
    if (w.blk_ints) (void)hipFree(w.blk_ints);
    w = Workspace();
}

#define TFS_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) { msg = std::string(#call) + " failed: " + hipGetErrorString(_e); return TF_ENODEVICE; } \
    } while (0)
#define TFS_BLAS(call)                                                                                  \
    do {                                                                                                \
        rocblas_status _s = (call);                                                                     \
        if (_s != rocblas_status_success) { msg = std::string(#call) + " failed (rocBLAS/rocSOLVER status " + std::to_string((int)_s) + ")"; return TF_ELINALG; } \
    } while (0)

inline int ensure(Workspace &w, int n, int n_mats, std::string &msg)
{
    if (!w.blas) {
        TFS_BLAS(rocblas_create_handle(&w.blas));
        TFS_HIP(hipMalloc((void **)&w.d_scal, 256 * sizeof(double)));   // [128, 256): scalar products of the DIIS history (64 per spin)
        TFS_HIP(hipMalloc((void **)&w.d_part, 256 * 8 * sizeof(double)));
        TFS_HIP(hipMalloc((void **)&w.d_info, sizeof(rocblas_int)));
        TFS_HIP(hipEventCreate(&w.ev0));
        TFS_HIP(hipEventCreate(&w.ev1));
    }
    const size_t need = (size_t)n_mats * n * n + 4 * (size_t)n;
    if (need > w.pool_doubles) {
        if (w.pool) (void)hipFree(w.pool);
        w.pool = nullptr; w.pool_doubles = 0;
        TFS_HIP(hipMalloc((void **)&w.pool, need * sizeof(double)));
        w.pool_doubles = need;
    }
    w.n = n;
    return TF_OK;
}

// ---- small element-wise kernels -------------------------------------------------------------------

__global__ void k_symmetrise(const double *__restrict__ in, double *__restrict__ out, int n)   // tuna_util.py:762
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    out[e] = (1.0 / 2.0) * (in[e] + in[(size_t)j * n + i]);
}

// F = H + J - 1/2 * hfx * K   (scf:525), then symmetrised by k_symmetrise
__global__ void k_fock(const double *__restrict__ H, const double *__restrict__ J, const double *__restrict__ K, double hfx,
                       double *__restrict__ F, int nn)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nn) F[e] = H[e] + J[e] - (1.0 / 2.0) * K[e] * hfx;
}

__global__ void k_axpby(double a, const double *__restrict__ x, double b, const double *__restrict__ y, double *__restrict__ out, int nn)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < nn) out[e] = a * x[e] + b * y[e];
}

__global__ void k_scale_cols(const double *__restrict__ V, const double *__restrict__ s, int mode, double *__restrict__ out, int n)
{
    // out[i][k] = V[i][k] * f(s[k]);  mode 0: s^-1/2, mode 1: 1/s     (row-major V, eigenvector k in column k)
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * n) return;
    const int k = e % n;
    out[e] = V[e] * (mode == 0 ? 1.0 / sqrt(s[k]) : 1.0 / s[k]);
}

struct Ptr8 { const double *p[8]; double c[8]; };
#define TF_MAX_DIIS 64           // history entries the native cycles hold (the reference keeps any `DIIS n`, scf:943-946; 8 per launch here)
__global__ void k_lincomb(Ptr8 a, int m, double *__restrict__ out, int nn, int accumulate)     // F_DIIS = sum_k c_k F_k (scf:1025)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nn) return;
    double s = accumulate ? out[e] : 0.0;
    for (int k = 0; k < m; ++k) s += a.c[k] * a.p[k][e];
    out[e] = s;
}

// out[k] = <x, a.p[k]> for k < m (m <= 8): all the scalar products of one SCF step in one launch and one read-back instead of one
// synchronising rocblas_ddot each (at N = 60 the cycle is bound by such round trips).  Single block, fixed summation tree.
__global__ void k_multi_dot(const double *__restrict__ x, Ptr8 a, int m, int nn, double *__restrict__ out)
{
    __shared__ double sm[8][256];
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    for (int e = threadIdx.x; e < nn; e += 256) {
        const double xv = x[e];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < m) acc[k] += xv * a.p[k][e];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st)
#pragma unroll
            for (int k = 0; k < 8; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < m) out[threadIdx.x] = sm[threadIdx.x][0];
}

// The same in two stages for long vectors (one block over 160 000 elements took 1.1 ms at N = 400, twice per SCF iteration):
// block b leaves its partial sums in part[b][0..7]; k_multi_dot_fin adds the blocks in index order (fixed: reproducible).
__global__ __launch_bounds__(256) void k_multi_dot_part(const double *__restrict__ x, Ptr8 a, int m, int nn, double *__restrict__ part)
{
    __shared__ double sm[8][256];
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    const int per = (nn + gridDim.x - 1) / gridDim.x, e0 = blockIdx.x * per, e1 = min(nn, e0 + per);
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
        const double xv = x[e];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < m) acc[k] += xv * a.p[k][e];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st)
#pragma unroll
            for (int k = 0; k < 8; ++k) sm[k][threadIdx.x] += sm[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 8) part[blockIdx.x * 8 + threadIdx.x] = sm[threadIdx.x][0];
}
__global__ void k_multi_dot_fin(const double *__restrict__ part, int nblk, int m, double *__restrict__ out)
{
    const int k = threadIdx.x;
    if (k >= m) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[b * 8 + k];
    out[k] = s;
}

// res[0] = max |a-b|, res[1] = sum (a-b)^2 in two stages (as above)
__global__ __launch_bounds__(256) void k_delta_norms_part(const double *__restrict__ a, const double *__restrict__ b, int nn, double *__restrict__ part)
{
    __shared__ double smax[256], ssum[256];
    double mx = 0.0, sm = 0.0;
    const int per = (nn + gridDim.x - 1) / gridDim.x, e0 = blockIdx.x * per, e1 = min(nn, e0 + per);
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
        const double d = a[e] - b[e];
        mx = fmax(mx, fabs(d));
        sm += d * d;
    }
    smax[threadIdx.x] = mx; ssum[threadIdx.x] = sm;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]); ssum[threadIdx.x] += ssum[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x * 8] = smax[0]; part[blockIdx.x * 8 + 1] = ssum[0]; }
}
__global__ void k_delta_norms_fin(const double *__restrict__ part, int nblk, double *__restrict__ res)
{
    if (threadIdx.x != 0) return;
    double mx = 0.0, sm = 0.0;
    for (int b = 0; b < nblk; ++b) { mx = fmax(mx, part[b * 8]); sm += part[b * 8 + 1]; }
    res[0] = mx; res[1] = sm;
}

// res[0] = max |a-b|, res[1] = sum (a-b)^2       (scf:285-286), single block
__global__ void k_delta_norms(const double *__restrict__ a, const double *__restrict__ b, int nn, double *__restrict__ res)
{
    __shared__ double smax[256], ssum[256];
    double mx = 0.0, sm = 0.0;
    for (int e = threadIdx.x; e < nn; e += 256) {
        const double d = a[e] - b[e];
        mx = fmax(mx, fabs(d));
        sm += d * d;
    }
    smax[threadIdx.x] = mx; ssum[threadIdx.x] = sm;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + s]); ssum[threadIdx.x] += ssum[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { res[0] = smax[0]; res[1] = ssum[0]; }
}

// Mulliken gross atomic populations: res[a] = sum_{i in atom a} (P S)_ii   (scf:789-818), single block
__global__ void k_mulliken(const double *__restrict__ P, const double *__restrict__ S, int n, int nA, double *__restrict__ res)
{
    __shared__ double s0[256], s1[256];
    double a0 = 0.0, a1 = 0.0;
    for (int e = threadIdx.x; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        const double v = P[e] * S[(size_t)j * n + i];
        if (i < nA) a0 += v; else a1 += v;
    }
    s0[threadIdx.x] = a0; s1[threadIdx.x] = a1;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { s0[threadIdx.x] += s0[threadIdx.x + s]; s1[threadIdx.x] += s1[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { res[0] = s0[0]; res[1] = s1[0]; }
}

// row-major C = alpha * op(A) * op(B) + beta * C, all n x n (k = inner dimension, default n)
inline void launch_multi_dot(Workspace &w, const double *x, const Ptr8 &a, int m, int nn, double *out)
{
    if (nn <= 8192) { hipLaunchKernelGGL(k_multi_dot, dim3(1), dim3(256), 0, TFS_ST, x, a, m, nn, out); return; }
    const int nblk = std::min(256, (nn + 2047) / 2048);
    hipLaunchKernelGGL(k_multi_dot_part, dim3(nblk), dim3(256), 0, TFS_ST, x, a, m, nn, w.d_part);
    hipLaunchKernelGGL(k_multi_dot_fin, dim3(1), dim3(64), 0, TFS_ST, w.d_part, nblk, m, out);
}
inline void launch_delta_norms(Workspace &w, const double *a, const double *b, int nn, double *res)
{
    if (nn <= 8192) { hipLaunchKernelGGL(k_delta_norms, dim3(1), dim3(256), 0, TFS_ST, a, b, nn, res); return; }
    const int nblk = std::min(256, (nn + 2047) / 2048);
    hipLaunchKernelGGL(k_delta_norms_part, dim3(nblk), dim3(256), 0, TFS_ST, a, b, nn, w.d_part);
    hipLaunchKernelGGL(k_delta_norms_fin, dim3(1), dim3(64), 0, TFS_ST, w.d_part, nblk, res);
}

inline rocblas_status gemm_rm(rocblas_handle h, bool tA, bool tB, int n, double alpha, const double *A, const double *B, double beta,
                              double *C)
{
    return rocblas_dgemm(h, tB ? rocblas_operation_transpose : rocblas_operation_none,
                         tA ? rocblas_operation_transpose : rocblas_operation_none, n, n, n, &alpha, B, n, A, n, &beta, C, n);
}

// ---- symmetry-blocked eigensolve ----------------------------------------------------------------------------------------------
// A diatomic on the z axis keeps the reflections x -> -x and y -> -y: every AO is even or odd under each (four classes, the same ones
// the packed tensor layout is blocked by), S, X = S^-1/2, the core Hamiltonian and a field along z do not connect different classes,
// and J, K of a class-diagonal density are class-diagonal again ((ij|kl) vanishes unless the four parities multiply to even,
// pyx:1324-1327).  The matrices the cycle diagonalises are then block diagonal after a permutation: at N = 400 blocks of
// 162 / 96 / 96 / 46 instead of 400, which rocsolver_dsyevd_strided_batched solves together in the time of the largest one
// (3.3 ms against 10.8 ms for the full matrix, `tools/gpu_eigh_blocks.py`).  Nothing is assumed: every call first measures the largest
// element that connects two classes (k_blk_cross) and declines -- the caller then solves the full matrix -- unless it is below
// 1e-14 of the largest element (a field along x, a symmetry-broken density or a basis without the structure end up there).
// Blocks are padded to the largest one with a decoupled diagonal JUST above the spectrum (1.01 x the Gershgorin bound: a padding value
// of n x max|a| -- 10^3 times the norm -- cost that factor in absolute accuracy, which the near-degenerate g/u pairs of a stretched
// diatomic cannot afford: 6e-7 in the core-guess orbital energies at N = 400); the eigenvalues of all blocks are ranked
// together (ties by block, then by position: a stable sort) and the vectors scattered back to the full basis.
inline void set_symmetry(Workspace &w, const std::vector<int> &cls) { if (cls != w.sym_cls) { w.sym_cls = cls; w.sym_n = 0; } }

// out[0] = max |a_ij|, out[1] = max |a_ij| over pairs of different classes, out[2] = max_i sum_j |a_ij| (Gershgorin: no eigenvalue lies
// outside [-out[2], out[2]]); one workgroup per row.  Non-negative doubles order like their bit patterns (a NaN ends up above
// everything: declined).
__global__ void k_blk_cross(const double *__restrict__ A, const int *__restrict__ cls, int n, unsigned long long *__restrict__ out)
{
    __shared__ double sa[256], sx[256], ss[256];
    const int i = blockIdx.x, ci = cls[i];
    double all = 0.0, cross = 0.0, sum = 0.0;
    for (int j = threadIdx.x; j < n; j += 256) {
        const double v = fabs(A[(size_t)i * n + j]);
        all = fmax(all, v);
        sum += v;
        if (cls[j] != ci) cross = fmax(cross, v);
    }
    sa[threadIdx.x] = all; sx[threadIdx.x] = cross; ss[threadIdx.x] = sum;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            sa[threadIdx.x] = fmax(sa[threadIdx.x], sa[threadIdx.x + st]); sx[threadIdx.x] = fmax(sx[threadIdx.x], sx[threadIdx.x + st]);
            ss[threadIdx.x] += ss[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(out, (unsigned long long)__double_as_longlong(sa[0]));
        atomicMax(out + 1, (unsigned long long)__double_as_longlong(sx[0]));
        atomicMax(out + 2, (unsigned long long)__double_as_longlong(ss[0]));
    }
}

// B[b][r][c] = A[idx[b][r]][idx[b][c]] (symmetrised).  sizes == nullptr: every block padded to mmax x mmax with a decoupled diagonal `big`
// (rocSOLVER's batched solver wants one size); sizes given: compact blocks, leading dimension sizes[b] (the batched Jacobi kernel)
__global__ void k_blk_gather(const double *__restrict__ A, const int *__restrict__ idx, int n, int mmax, double big, double *__restrict__ B,
                             const int *__restrict__ sizes)
{
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= mmax * mmax) return;
    const int r = e / mmax, c = e - r * mmax;
    const int i = idx[b * mmax + r], j = idx[b * mmax + c];
    if (sizes) {
        if (i >= 0 && j >= 0) B[(size_t)b * mmax * mmax + (size_t)r * sizes[b] + c] = 0.5 * (A[(size_t)i * n + j] + A[(size_t)j * n + i]);
        return;
    }
    double v;
    if (i >= 0 && j >= 0) v = 0.5 * (A[(size_t)i * n + j] + A[(size_t)j * n + i]);
    else v = (r == c) ? big : 0.0;
    B[((size_t)b * mmax + r) * mmax + c] = v;
}

// global ranks of the eigenvalues of all blocks (padding excluded): vals[rank] = value, src[rank] = b * mmax + t; single block of 1024
__global__ void k_blk_rank(const double *__restrict__ D, const int *__restrict__ idx, int nb, int mmax, double *__restrict__ vals, int *__restrict__ src)
{
    // member t of block b is real iff idx[b][t] >= 0 (members come first); the padded problems return their values ascending, so the
    // real ones are the first m_b of each block
    const int tot = nb * mmax;
    for (int q = threadIdx.x; q < tot; q += blockDim.x) {
        if (idx[q] < 0) continue;
        const double d = D[q];
        int rank = 0;
        for (int p = 0; p < tot; ++p) {
            if (idx[p] < 0) continue;
            const double dp = D[p];
            rank += (dp < d || (dp == d && p < q)) ? 1 : 0;
        }
        vals[rank] = d;
        src[rank] = q;
    }
}

// W[rank][:] = eigenvector `src[rank]` of its block, scattered to the full basis (zero outside the block); one workgroup per row
__global__ void k_blk_scatter(const double *__restrict__ B, const int *__restrict__ idx, const int *__restrict__ src, int n, int mmax,
                              double *__restrict__ W, const int *__restrict__ sizes)
{
    const int r = blockIdx.x;
    const int q = src[r], b = q / mmax, t = q - b * mmax;
    double *row = W + (size_t)r * n;
    for (int c = threadIdx.x; c < n; c += blockDim.x) row[c] = 0.0;
    __syncthreads();
    const double *v = B + (size_t)b * mmax * mmax + (size_t)t * (sizes ? sizes[b] : mmax);   // row t of the block in row-major terms = column t for rocSOLVER
    for (int c = threadIdx.x; c < mmax; c += blockDim.x) {
        const int i = idx[b * mmax + c];
        if (i >= 0) row[i] = v[c];
    }
}

// block of the eigenvector of global rank r
__global__ void k_blk_labels(const int *__restrict__ src, int mmax, int n, int *__restrict__ out)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = src[r] / mmax;
}

// TF_OK: solved (W rows = eigenvectors, vals ascending).  TF_EINVAL with an empty msg: declined, the caller solves the full matrix.
inline int eigh_blocked(Workspace &w, int n, double *W, double *vals, std::string &msg)
{
    static const bool off = getenv("TF_EIGH_BLOCKS") && getenv("TF_EIGH_BLOCKS")[0] == '0';
    static const int nmin = getenv("TF_EIGH_BLOCKS_NMIN") ? atoi(getenv("TF_EIGH_BLOCKS_NMIN")) : 40;   // below: one in-LDS Jacobi of the whole matrix is as fast
    if (off || (int)w.sym_cls.size() != n || n < nmin) return TF_EINVAL;
    if (w.sym_n != n) {                                            // tables for this class vector
        int m[4] = {0, 0, 0, 0};
        for (int c : w.sym_cls) { if (c < 0 || c > 3) return TF_EINVAL; ++m[c]; }
        int nb = 0, mmax = 0, blk_of[4] = {-1, -1, -1, -1};
        for (int c = 0; c < 4; ++c) if (m[c] > 0) { blk_of[c] = nb; w.sym_m[nb] = m[c]; ++nb; mmax = std::max(mmax, m[c]); }
        w.sym_nb = nb; w.sym_mmax = mmax; w.sym_n = n;
        if (w.sym_idx) (void)hipFree(w.sym_idx);
        if (w.sym_buf) (void)hipFree(w.sym_buf);
        if (w.sym_src) (void)hipFree(w.sym_src);
        w.sym_idx = nullptr; w.sym_buf = nullptr; w.sym_src = nullptr;
        std::vector<int> idx((size_t)4 * mmax + n + 4, -1);         // members of the blocks | class of every function | block sizes
        int fill[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) { const int b = blk_of[w.sym_cls[i]]; idx[(size_t)b * mmax + fill[b]++] = i; idx[(size_t)4 * mmax + i] = w.sym_cls[i]; }
        for (int b = 0; b < 4; ++b) idx[(size_t)4 * mmax + n + b] = b < nb ? w.sym_m[b] : 0;
        w.sym_prev_n = 0; w.blk_ref_n = 0; w.blk_alt_ref_n = 0;
        if (w.sym_prev) { (void)hipFree(w.sym_prev); w.sym_prev = nullptr; }
        if (w.blk_X) { (void)hipFree(w.blk_X); w.blk_X = nullptr; }
        if (w.blk_alt_X) { (void)hipFree(w.blk_alt_X); w.blk_alt_X = nullptr; }
        if (mmax <= 64) TFS_HIP(hipMalloc((void **)&w.sym_prev, (size_t)nb * mmax * mmax * sizeof(double)));
        TFS_HIP(hipMalloc((void **)&w.sym_idx, idx.size() * sizeof(int)));
        TFS_HIP(tfs_memcpy(w.sym_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
        TFS_HIP(hipMalloc((void **)&w.sym_buf, ((size_t)nb * mmax * mmax + 2 * (size_t)nb * mmax + 4) * sizeof(double)));
        TFS_HIP(hipMalloc((void **)&w.sym_src, (size_t)n * sizeof(int)));
        if (!w.sym_info) TFS_HIP(hipMalloc((void **)&w.sym_info, 4 * sizeof(rocblas_int)));
    }
    const int nb = w.sym_nb, mmax = w.sym_mmax;
    if (nb < 2 || 4 * mmax > 3 * n) return TF_EINVAL;             // one class holds (nearly) everything: nothing to gain
    double *B = w.sym_buf, *D = B + (size_t)nb * mmax * mmax, *E = D + (size_t)nb * mmax;
    unsigned long long *flag = (unsigned long long *)(E + (size_t)nb * mmax);
    const int *cls = w.sym_idx + (size_t)4 * mmax;
    TFS_HIP(hipMemsetAsync(flag, 0, 3 * sizeof(double), TFS_ST));
    hipLaunchKernelGGL(k_blk_cross, dim3(n), dim3(256), 0, TFS_ST, W, cls, n, flag);
    double h[3];
    TFS_HIP(tfs_memcpy(h, flag, sizeof(h), hipMemcpyDeviceToHost));
    if (!(h[1] <= 1e-14 * h[0]) || !std::isfinite(h[0]) || !std::isfinite(h[2])) { ++w.sym_declined; return TF_EINVAL; }
    const int g = (mmax * mmax + 255) / 256;
    const int *sizes = nullptr;
    if (mmax <= 64) {
        // every block fits the in-LDS Jacobi kernel (tf_jacobi.hip.h): one launch, a workgroup per block, compact blocks, warm-started
        // from the block eigenvectors of the previous solve of the cycle (N2/cc-pVTZ: 60 = 26 + 14 + 14 + 6)
        static const bool no_warm = getenv("TF_EIGH_COLD") != nullptr;
        sizes = w.sym_idx + (size_t)4 * mmax + n;
        hipLaunchKernelGGL(k_blk_gather, dim3(g, nb), dim3(256), 0, TFS_ST, W, w.sym_idx, n, mmax, 0.0, B, sizes);
        const double *V0 = (w.warm_ok && !no_warm && w.sym_prev_n == n) ? w.sym_prev : nullptr;
        hipError_t e = hipSuccess;
        if (!tfjac::launch_batch(nb, mmax, sizes, B, (long long)mmax * mmax, D, mmax, (int *)w.sym_info, TFS_ST, &e, V0, w.warm_ok ? w.sym_prev : nullptr)) {
            msg = std::string("batched Jacobi eigensolver launch failed: ") + hipGetErrorString(e);
            return TF_ENODEVICE;
        }
        w.sym_prev_n = w.warm_ok ? n : 0;
    } else {
        hipLaunchKernelGGL(k_blk_gather, dim3(g, nb), dim3(256), 0, TFS_ST, W, w.sym_idx, n, mmax, 1.01 * h[2] + 1e-300, B, sizes);
        TFS_BLAS(rocsolver_dsyevd_strided_batched(w.blas, rocblas_evect_original, rocblas_fill_upper, mmax, B, mmax, (rocblas_stride)mmax * mmax,
                                                  D, mmax, E, mmax, w.sym_info, nb));
    }
    hipLaunchKernelGGL(k_blk_rank, dim3(1), dim3(1024), 0, TFS_ST, D, w.sym_idx, nb, mmax, vals, w.sym_src);
    hipLaunchKernelGGL(k_blk_scatter, dim3(n), dim3(128), 0, TFS_ST, B, w.sym_idx, w.sym_src, n, mmax, W, sizes);
    ++w.sym_solves;
    w.sym_last_blocked = true;
    return TF_OK;
}

// Symmetric eigenproblem: W (in: symmetric matrix, out: row k = eigenvector k in row-major terms), vals ascending.
// n <= 64: single-launch in-LDS Jacobi (tf_jacobi.hip.h; measured 0.06/0.23/0.96 ms at n = 10/28/60 against 0.24/0.67/1.22 ms
// for dsyevd); larger: rocsolver_dsyevd (faster from n ~ 70 on).  TF_EIGH=rocsolver|jacobi overrides.
inline int eigh(Workspace &w, int n, double *W, double *vals, double *work_e, std::string &msg)
{
    static const bool force_rocsolver = getenv("TF_EIGH") && std::string(getenv("TF_EIGH")) == "rocsolver";
    static const bool force_jacobi = getenv("TF_EIGH") && std::string(getenv("TF_EIGH")) == "jacobi";
    static const int jac_nmax = getenv("TF_JACOBI_NMAX") ? std::min(TFJ_NMAX, std::max(2, atoi(getenv("TF_JACOBI_NMAX")))) : 64;
    w.sym_last_blocked = false;
    if (!force_rocsolver && !force_jacobi) {                       // the symmetry blocks of a diatomic, solved together (declines if there are none)
        std::string bmsg;
        const int rb = eigh_blocked(w, n, W, vals, bmsg);
        if (rb == TF_OK) return TF_OK;
        if (!bmsg.empty()) { msg = bmsg; return rb; }
    }
    if (!force_rocsolver && n >= 2 && n <= (force_jacobi ? TFJ_NMAX : jac_nmax)) {
        if (w.jac_cap < (size_t)n * n) {
            if (w.jac_scratch) (void)hipFree(w.jac_scratch);
            w.jac_scratch = nullptr; w.jac_cap = 0;
            TFS_HIP(hipMalloc((void **)&w.jac_scratch, (size_t)n * n * sizeof(double)));
            w.jac_cap = (size_t)n * n;
        }
        if (w.jac_prev_cap < (size_t)n * n) {
            if (w.jac_prev) (void)hipFree(w.jac_prev);
            w.jac_prev = nullptr; w.jac_prev_cap = 0; w.jac_prev_n = 0;
            TFS_HIP(hipMalloc((void **)&w.jac_prev, (size_t)n * n * sizeof(double)));
            w.jac_prev_cap = (size_t)n * n;
        }
        static const bool no_warm = getenv("TF_EIGH_COLD") != nullptr;
        const double *V0 = (w.warm_ok && !no_warm && w.jac_prev_n == n) ? w.jac_prev : nullptr;
        hipError_t e = hipSuccess;
        if (tfjac::launch(n, W, vals, w.jac_scratch, (int *)w.d_info, TFS_ST, &e, V0, w.warm_ok ? w.jac_prev : nullptr)) {
            if (w.warm_ok) w.jac_prev_n = n;
            return TF_OK;
        }
        if (e != hipSuccess) { msg = std::string("Jacobi eigensolver launch failed: ") + hipGetErrorString(e); return TF_ENODEVICE; }
    }
    TFS_BLAS(rocsolver_dsyevd(w.blas, rocblas_evect_original, rocblas_fill_upper, n, W, n, vals, work_e, w.d_info));
    return TF_OK;
}

// dense solve A x = b for the (m <= 9) DIIS system; false if A is exactly singular (np.linalg.solve -> LinAlgError)
// ---- warm-started eigenvector refinement ------------------------------------------------------------------------------------
// Inside an SCF cycle successive Fock matrices differ little, and what the cycle needs from a diagonalisation is the projector on
// the n_occ lowest eigenvectors (scf:183-211, 222-250).  Given the eigenvectors X of the previous solve, one step of the
// first-order refinement of Ogita & Aishima (Japan J. Indust. Appl. Math. 35 (2018) 1007), in the form used here:
//     X <- X (3/2 I - 1/2 X^T X)                     (Newton-Schulz: orthonormal to the square of the previous defect)
//     S = X^T A X,  lambda_i = s_ii,  e_ij = s_ij / (lambda_j - lambda_i),  X <- X + X E
// is five n^3 GEMMs on the matrix cores and converges quadratically.  Every occupied-virtual pair is rotated (its denominator is at
// least the gap); a pair inside the occupied or inside the virtual space is rotated only where that is well conditioned
// (|s_ij| <= 0.05 |lambda_j - lambda_i| and the difference above rounding level) -- the rotation inside a (near-)degenerate cluster
// does not change the projector, and diatomics are full of exact degeneracies (pi, delta, g/u pairs of separated atoms).
// Converged when the largest occupied-virtual rotation is below 1e-9.  No convergence in 16 steps, a rotation above 0.3 or a closed
// gap: the caller diagonalises (rocsolver_dsyevd) and restarts from those vectors.  At n = 400 a step costs ~0.15 ms against 9 ms
// for dsyevd (`tools/gpu_eigh_probe.py`); orbitals and orbital energies for the caller are produced once at the end of the cycle by a
// real eigensolve.  Storage: rows of Xr are the eigenvectors (Xr = X^T).

// lam[i] = s_ii, occupation weights w[i] (1 for the n_occ lowest), scal = {homo, lumo}; single block
__global__ void k_ref_diag(const double *__restrict__ S, int n, int n_occ, double *__restrict__ lam, double *__restrict__ wocc,
                           double *__restrict__ scal)
{
    __shared__ double s0[1024], s1[1024];
    for (int i = threadIdx.x; i < n; i += 1024) lam[i] = S[(size_t)i * n + i];
    __syncthreads();
    // ranks: the n_occ lowest values are occupied (ties broken by index, as a stable sort would)
    double homo = -1e300, lumo = 1e300;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double li = lam[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) { const double lj = lam[j]; rank += (lj < li || (lj == li && j < i)) ? 1 : 0; }
        const bool occ = rank < n_occ;
        wocc[i] = occ ? 1.0 : 0.0;
        if (occ) homo = fmax(homo, li); else lumo = fmin(lumo, li);
    }
    s0[threadIdx.x] = homo; s1[threadIdx.x] = lumo;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
        if (threadIdx.x < st) { s0[threadIdx.x] = fmax(s0[threadIdx.x], s0[threadIdx.x + st]); s1[threadIdx.x] = fmin(s1[threadIdx.x], s1[threadIdx.x + st]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { scal[0] = s0[0]; scal[1] = s1[0]; }
}

// E (row-major, antisymmetric) and per block: max |e_ij| over all pairs, and over the occupied-virtual pairs
// vcls (optional): block label of every vector (the exact solve they come from went block by block over the parity classes).  Vectors
// of different blocks are never rotated into each other: what couples them is rounding residue of the class-diagonal problem (the
// blocked solver dropped the same elements), and vectors that stay class-pure give densities with exact zeros between the classes,
// which the Fock build then exploits (tf_device.hip: the class-diagonal task list).  A coupling that is NOT residue (> 1e-6 as a rotation)
// is reported as a rotation of 1: the caller falls back to an exact solve, which will decline the blocks.
__global__ void k_ref_E(const double *__restrict__ S, const double *__restrict__ lam, const double *__restrict__ wocc, int n,
                        double *__restrict__ E, double *__restrict__ blockmax, const int *__restrict__ vcls)
{
    __shared__ double sm[256], so[256];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    bool ov = false, cross_bad = false;
    if (e < n * n) {
        const int i = e / n, j = e - i * n;
        if (i != j) {
            const double sij = 0.5 * (S[e] + S[(size_t)j * n + i]);
            const double dl = lam[j] - lam[i];
            ov = wocc[i] != wocc[j];
            if (ov || (fabs(sij) <= TF_REF_INTRA * fabs(dl) && fabs(dl) > TF_REF_CLUSTER)) v = sij / dl;
            if (vcls && vcls[i] != vcls[j]) {
                cross_bad = ov && fabs(v) > 1e-6;
                v = 0.0;
            }
        }
        E[e] = v;
    }
    sm[threadIdx.x] = cross_bad ? 1.0 : fabs(v); so[threadIdx.x] = cross_bad ? 1.0 : (ov ? fabs(v) : 0.0);
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) { sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + st]); so[threadIdx.x] = fmax(so[threadIdx.x], so[threadIdx.x + st]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) { blockmax[2 * blockIdx.x] = sm[0]; blockmax[2 * blockIdx.x + 1] = so[0]; }
}

// out[i][:] = w[i] * X[i][:]
__global__ void k_scale_rows(const double *__restrict__ X, const double *__restrict__ w, double *__restrict__ out, int n)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * n) out[e] = w[e / n] * X[e];
}

inline int ref_ensure(Workspace &w, int n, std::string &msg)
{
    const size_t need = 6 * (size_t)n * n + 3 * (size_t)n + 2 * (((size_t)n * n + 255) / 256) + 8;   // (+ n: block labels of the vectors)
    if (need > w.ref_cap) {
        if (w.ref_buf) (void)hipFree(w.ref_buf);
        w.ref_buf = nullptr; w.ref_cap = 0; w.ref_n = 0;
        TFS_HIP(hipMalloc((void **)&w.ref_buf, need * sizeof(double)));
        w.ref_cap = need;
    }
    const size_t npin = 2 * (((size_t)n * n + 255) / 256) + 2;
    if (npin > w.h_pin_cap) {
        if (w.h_pin) (void)hipHostFree(w.h_pin);
        w.h_pin = nullptr; w.h_pin_cap = 0;
        TFS_HIP(hipHostMalloc((void **)&w.h_pin, npin * sizeof(double), hipHostMallocDefault));
        w.h_pin_cap = npin;
    }
    if (!w.ev_pin) TFS_HIP(hipEventCreateWithFlags(&w.ev_pin, hipEventDisableTiming));
    return TF_OK;
}

// Refines the stored vectors against the symmetric A (device, row-major n x n).  TF_OK: *Xocc (a buffer of the workspace) holds the
// rows of the n_occ lowest eigenvectors (other rows zero).  TF_ELINALG: no convergence -- the caller diagonalises and calls ref_store.
inline int ref_refine(Workspace &w, int n, int n_occ, const double *A, double **Xocc, std::string &msg)
{
    if (w.ref_n != n || !w.ref_buf) return TF_ELINALG;
    static const bool dbg = getenv("TF_DEBUG") != nullptr;
    const size_t nn = (size_t)n * n;
    double *X = w.ref_buf, *Xn = X + nn, *G = Xn + nn, *Y = G + nn, *S = Y + nn, *E = S + nn;
    double *lam = E + nn, *wocc = lam + n, *bmax = wocc + n;
    const int g = (int)((nn + 255) / 256);
    const int *vcls = (w.ref_vcls_n == n) ? reinterpret_cast<const int *>(w.ref_buf + 6 * nn + 2 * (size_t)n + 2 * (size_t)g + 8) : nullptr;
    // X <- (3/2 I - 1/2 G) X with G = X X^T (rows are the vectors)
    auto orthonormalise = [&]() -> int {
        TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, X, X, 0.0, G));
        TFS_HIP(hipMemcpyAsync(Xn, X, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        TFS_BLAS(gemm_rm(w.blas, false, false, n, -0.5, G, X, 1.5, Xn));
        std::swap(X, Xn);
        return TF_OK;
    };
    ++w.ref_solves;
    int rc = TF_OK;
    bool ok = false;
    // Every step ends with the update X <- X + E^T X and the orthonormalisation that either the next step or the end of the solve
    // needs; they are queued BEFORE the host looks at the step's convergence numbers (asynchronous copy to pinned memory + event), so
    // the read-back costs no idle time on the device.  (The stored vectors are orthonormal: no orthonormalisation before step 0.)
    const size_t npin = 2 * (size_t)g + 2;
    if (!w.h_pin || w.h_pin_cap < npin || !w.ev_pin) return TF_ELINALG;
    for (int step = 0; step < 16 && !ok; ++step) {
        TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, X, A, 0.0, Y));         // rows A x_i
        TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, Y, X, 0.0, S));          // S = X^T A X
        hipLaunchKernelGGL(k_ref_diag, dim3(1), dim3(1024), 0, TFS_ST, S, n, n_occ, lam, wocc, bmax + 2 * (size_t)g);
        hipLaunchKernelGGL(k_ref_E, dim3(g), dim3(256), 0, TFS_ST, S, lam, wocc, n, E, bmax, vcls);
        TFS_HIP(hipMemcpyAsync(w.h_pin, bmax, npin * sizeof(double), hipMemcpyDeviceToHost, TFS_ST));
        TFS_HIP(hipEventRecord(w.ev_pin, TFS_ST));
        TFS_HIP(hipMemcpyAsync(Xn, X, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        TFS_BLAS(gemm_rm(w.blas, true, false, n, 1.0, E, X, 1.0, Xn));         // X + E^T X   (rows)
        std::swap(X, Xn);
        if ((rc = orthonormalise())) return rc;
        TFS_HIP(hipEventSynchronize(w.ev_pin));
        const double *hb = w.h_pin;
        const double h[2] = {hb[2 * (size_t)g], hb[2 * (size_t)g + 1]};
        double emax = 0.0, eov = 0.0;
        for (int b = 0; b < g; ++b) { emax = std::max(emax, hb[2 * b]); eov = std::max(eov, hb[2 * b + 1]); }
        if (dbg) fprintf(stderr, "[tf refine] step %d: homo %.6f lumo %.6f max|E| %.3e max|E_ov| %.3e\n", step, h[0], h[1], emax, eov);
        if (!std::isfinite(emax) || !(h[1] > h[0]) || emax > 0.3) break;       // (the queued update is discarded with the vectors)
        w.ref_steps += 1;
        ok = eov < 1e-9 && emax < 0.1;       // rotations inside the occupied or the virtual space leave the projector alone
    }
    if (ok) {
        if (X != w.ref_buf) {                                                  // keep the vectors in the first slot for the next solve
            TFS_HIP(hipMemcpyAsync(w.ref_buf, X, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
            X = w.ref_buf;
        }
        double *Xw = w.ref_buf + 2 * nn;                                       // G's slot: free now
        hipLaunchKernelGGL(k_scale_rows, dim3(g), dim3(256), 0, TFS_ST, X, wocc, Xw, n);
        *Xocc = Xw;
        return TF_OK;
    }
    ++w.ref_fallbacks;
    w.ref_n = 0;
    return TF_ELINALG;
}

// n <= 64: the same refinement in one launch with every matrix in LDS (tf_refine.hip.h); one 8-byte status read per solve.
inline int ref_refine_lds(Workspace &w, int n, int n_occ, const double *A, double **Xocc, std::string &msg)
{
    if (w.ref_n != n || !w.ref_buf || n > TFR_NMAX) return TF_ELINALG;
    const size_t nn = (size_t)n * n;
    double *X = w.ref_buf, *Xw = X + 2 * nn, *lam = X + 6 * nn, *wocc = lam + n;
    int *status = reinterpret_cast<int *>(w.d_scal + 62);
    hipError_t e = hipSuccess;
    ++w.ref_solves;
    if (!tfref::launch(n, n_occ, A, X, lam, wocc, status, TFS_ST, &e)) {
        if (e != hipSuccess) { msg = std::string("refinement kernel launch failed: ") + hipGetErrorString(e); return TF_ENODEVICE; }
        return TF_ELINALG;
    }
    int h[2] = {0, 0};
    TFS_HIP(tfs_memcpy(h, status, 2 * sizeof(int), hipMemcpyDeviceToHost));
    static const bool dbg = getenv("TF_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[tf refine/lds] n %d: %s after %d steps\n", n, h[0] ? "converged" : "NOT converged", h[1]);
    if (!h[0]) { ++w.ref_fallbacks; w.ref_n = 0; return TF_ELINALG; }
    w.ref_steps += h[1];
    const int g = (int)((nn + 255) / 256);
    hipLaunchKernelGGL(k_scale_rows, dim3(g), dim3(256), 0, TFS_ST, X, wocc, Xw, n);
    *Xocc = Xw;
    return TF_OK;
}

// n <= 64, fused: A = sym(Xo^T Fao Xo), the refinement and the density P = occ * sym(C_occ C_occ^T) in the same launch
// (tf_refine.hip.h); nothing else of the "diagonalise and rebuild the density" step is left to launch.
inline int ref_density_lds(Workspace &w, int n, int n_occ, const double *Fao, const double *Xo, double *Pout, double occ, std::string &msg)
{
    if (w.ref_n != n || !w.ref_buf || n > TFR_NMAX) return TF_ELINALG;
    const size_t nn = (size_t)n * n;
    double *X = w.ref_buf, *lam = X + 6 * nn, *wocc = lam + n;
    int *status = reinterpret_cast<int *>(w.d_scal + 62);
    hipError_t e = hipSuccess;
    ++w.ref_solves;
    if (!tfref::launch(n, n_occ, nullptr, X, lam, wocc, status, TFS_ST, &e, Fao, Xo, Pout, occ)) {
        if (e != hipSuccess) { msg = std::string("refinement kernel launch failed: ") + hipGetErrorString(e); return TF_ENODEVICE; }
        return TF_ELINALG;
    }
    int h[2] = {0, 0};
    TFS_HIP(tfs_memcpy(h, status, 2 * sizeof(int), hipMemcpyDeviceToHost));
    static const bool dbg = getenv("TF_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[tf refine/lds fused] n %d: %s after %d steps\n", n, h[0] ? "converged" : "NOT converged", h[1]);
    if (!h[0]) { ++w.ref_fallbacks; w.ref_n = 0; return TF_ELINALG; }
    w.ref_steps += h[1];
    return TF_OK;
}

// after a real eigensolve: rows of V (row-major, as eigh() leaves them) become the refinement start
// ---- refinement block by block (64 < n, every parity block <= 64 vectors: cc-pVQZ-size molecules) ---------------------------------
// The exact solves of such a matrix already run as one batched launch of the in-LDS Jacobi kernel over its blocks (eigh_blocked); the
// refinement does the same with the in-LDS refinement kernel (tf_refine.hip.h): a workgroup per block, its vectors and its block of A in
// LDS, the five products of a step on the matrix core -- one launch and one status read per solve instead of five rocBLAS GEMMs, two
// kernels and a read-back per STEP at the full dimension.  Each block keeps the number of occupied vectors the last exact solve gave it;
// after the launch the global aufbau order is verified (highest occupied value of all blocks below the lowest empty one), else -- or if
// a block did not converge, or A has an element between two classes -- the caller diagonalises.
__global__ void k_blk_noccs(const int *__restrict__ src, int mmax, int n_occ, int *__restrict__ noccs)
{
    __shared__ int cnt[4];
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < n_occ; r += blockDim.x) atomicAdd(&cnt[src[r] / mmax], 1);
    __syncthreads();
    if (threadIdx.x < 4) noccs[threadIdx.x] = cnt[threadIdx.x];
}

// combined[0] = 1 iff every block converged and the occupied values of all blocks lie below all empty ones; combined[1] = most steps
__global__ void k_blk_ref_finish(const int *__restrict__ sizes, const int *__restrict__ status, const double *__restrict__ lam,
                                 const double *__restrict__ wocc, int nb, int mmax, int *__restrict__ combined)
{
    __shared__ double sh[256], sl[256];
    double homo = -1e300, lumo = 1e300;
    for (int q = threadIdx.x; q < nb * mmax; q += 256) {
        const int b = q / mmax, t = q - b * mmax;
        if (t >= sizes[b]) continue;
        if (wocc[q] != 0.0) homo = fmax(homo, lam[q]); else lumo = fmin(lumo, lam[q]);
    }
    sh[threadIdx.x] = homo; sl[threadIdx.x] = lumo;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) { sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + st]); sl[threadIdx.x] = fmin(sl[threadIdx.x], sl[threadIdx.x + st]); }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int ok = (sh[0] < sl[0]) ? 1 : 0, steps = 0;
        for (int b = 0; b < nb; ++b) { ok = ok && status[2 * b]; steps = max(steps, status[2 * b + 1]); }
        combined[0] = ok; combined[1] = steps;
    }
}

// rows of the occupied vectors of all blocks in the full basis (block after block; the others zero): what ref_refine hands back
__global__ void k_blk_xocc(const double *__restrict__ Xb, const double *__restrict__ wocc, const int *__restrict__ idx, const int *__restrict__ sizes,
                           int n, int mmax, double *__restrict__ out)
{
    int r = blockIdx.x, b = 0;
    while (b < 3 && r >= sizes[b]) { r -= sizes[b]; ++b; }
    double *row = out + (size_t)blockIdx.x * n;
    for (int c = threadIdx.x; c < n; c += blockDim.x) row[c] = 0.0;
    __syncthreads();
    const double wv = wocc[b * mmax + r];
    if (wv == 0.0) return;
    const int m = sizes[b];
    const double *v = Xb + (size_t)b * mmax * mmax + (size_t)r * m;
    for (int c = threadIdx.x; c < m; c += blockDim.x) row[idx[b * mmax + c]] = wv * v[c];
}

inline int ref_refine_blocks(Workspace &w, int n, const double *A, double **Xocc, std::string &msg)
{
    if (w.blk_ref_n != n || !w.blk_X || w.sym_n != n || w.sym_mmax > TFR_NMAX || !w.ref_buf) return TF_ELINALG;
    const int nb = w.sym_nb, mmax = w.sym_mmax;
    double *B = w.sym_buf, *D = B + (size_t)nb * mmax * mmax, *E = D + (size_t)nb * mmax;
    unsigned long long *flag = (unsigned long long *)(E + (size_t)nb * mmax);
    const int *cls = w.sym_idx + (size_t)4 * mmax, *sizes = cls + n;
    int *noccs = w.blk_ints + w.blk_nocc_off, *status = w.blk_ints + 4, *combined = w.blk_ints + 12;
    ++w.ref_solves; ++w.blk_solves;
    TFS_HIP(hipMemsetAsync(flag, 0, 3 * sizeof(double), TFS_ST));
    hipLaunchKernelGGL(k_blk_cross, dim3(n), dim3(256), 0, TFS_ST, A, cls, n, flag);
    hipLaunchKernelGGL(k_blk_gather, dim3((mmax * mmax + 255) / 256, nb), dim3(256), 0, TFS_ST, A, w.sym_idx, n, mmax, 0.0, B, sizes);
    hipError_t e = hipSuccess;
    if (!tfref::launch_batch(nb, mmax, sizes, noccs, B, w.blk_X, (long long)mmax * mmax, D, E, mmax, status, TFS_ST, &e)) {
        if (e != hipSuccess) { msg = std::string("blocked refinement kernel launch failed: ") + hipGetErrorString(e); return TF_ENODEVICE; }
        return TF_ELINALG;
    }
    hipLaunchKernelGGL(k_blk_ref_finish, dim3(1), dim3(256), 0, TFS_ST, sizes, status, D, E, nb, mmax, combined);
    double *Xw = w.ref_buf + 2 * (size_t)n * n;
    hipLaunchKernelGGL(k_blk_xocc, dim3(n), dim3(64), 0, TFS_ST, w.blk_X, E, w.sym_idx, sizes, n, mmax, Xw);
    int h[2] = {0, 0};
    double hf[3] = {0.0, 1.0, 0.0};
    TFS_HIP(tfs_memcpy(h, combined, sizeof(h), hipMemcpyDeviceToHost));
    TFS_HIP(tfs_memcpy(hf, flag, sizeof(hf), hipMemcpyDeviceToHost));
    static const bool dbg = getenv("TF_DEBUG") != nullptr;
    const bool diag_ok = std::isfinite(hf[0]) && hf[1] <= 1e-14 * hf[0];
    if (dbg) fprintf(stderr, "[tf refine/blocks] n %d (%d blocks <= %d): %s after %d steps%s\n", n, nb, mmax, h[0] ? "converged" : "NOT converged", h[1],
                     diag_ok ? "" : ", matrix not class-diagonal");
    if (!h[0] || !diag_ok) { ++w.ref_fallbacks; ++w.blk_fallbacks; w.blk_ref_n = 0; w.ref_n = 0; return TF_ELINALG; }
    w.ref_steps += h[1];
    *Xocc = Xw;
    return TF_OK;
}

inline int ref_store(Workspace &w, int n, const double *V, std::string &msg, int n_occ = -1)
{
    int rc = ref_ensure(w, n, msg);
    if (rc) return rc;
    TFS_HIP(hipMemcpyAsync(w.ref_buf, V, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
    w.ref_n = n;
    w.ref_vcls_n = 0;
    if (w.sym_last_blocked && w.sym_n == n && w.sym_src) {         // V are the vectors of the blocked solve just done: keep their block labels
        const size_t nn = (size_t)n * n, g = (nn + 255) / 256;
        int *vcls = reinterpret_cast<int *>(w.ref_buf + 6 * nn + 2 * (size_t)n + 2 * g + 8);
        hipLaunchKernelGGL(k_blk_labels, dim3((n + 255) / 256), dim3(256), 0, TFS_ST, w.sym_src, w.sym_mmax, n, vcls);
        w.ref_vcls_n = n;
    }
    // every block fits the in-LDS refinement kernel: keep the blocks' vectors (still in the solver's buffer) and their occupations
    static const bool blk_off = getenv("TF_REFINE_BLOCKS") && getenv("TF_REFINE_BLOCKS")[0] == '0';
    w.blk_ref_n = 0;
    if (!blk_off && n_occ > 0 && n > TFR_NMAX && w.sym_last_blocked && w.sym_n == n && w.sym_mmax <= TFR_NMAX && w.sym_src) {
        const size_t bytes = (size_t)w.sym_nb * w.sym_mmax * w.sym_mmax * sizeof(double);
        if (!w.blk_X) TFS_HIP(hipMalloc((void **)&w.blk_X, bytes));
        if (!w.blk_ints) TFS_HIP(hipMalloc((void **)&w.blk_ints, 32 * sizeof(int)));
        TFS_HIP(hipMemcpyAsync(w.blk_X, w.sym_buf, bytes, hipMemcpyDeviceToDevice, TFS_ST));
        hipLaunchKernelGGL(k_blk_noccs, dim3(1), dim3(64), 0, TFS_ST, w.sym_src, w.sym_mmax, n_occ, w.blk_ints + w.blk_nocc_off);
        w.blk_ref_n = n;
    }
    return TF_OK;
}

inline bool small_solve(int m, std::vector<double> A, std::vector<double> b, std::vector<double> &x)
{
    for (int c = 0; c < m; ++c) {
        int piv = c;
        for (int r = c + 1; r < m; ++r)
            if (std::fabs(A[r * m + c]) > std::fabs(A[piv * m + c])) piv = r;
        if (A[piv * m + c] == 0.0 || !std::isfinite(A[piv * m + c])) return false;
        if (piv != c) {
            for (int k = 0; k < m; ++k) std::swap(A[c * m + k], A[piv * m + k]);
            std::swap(b[c], b[piv]);
        }
        for (int r = c + 1; r < m; ++r) {
            const double f = A[r * m + c] / A[c * m + c];
            if (f == 0.0) continue;
            for (int k = c; k < m; ++k) A[r * m + k] -= f * A[c * m + k];
            b[r] -= f * b[c];
        }
    }
    x.assign(m, 0.0);
    for (int r = m - 1; r >= 0; --r) {
        double s = b[r];
        for (int k = r + 1; k < m; ++k) s -= A[r * m + k] * x[k];
        x[r] = s / A[r * m + r];
    }
    return true;
}

// X = S^-1/2, S^-1, smallest eigenvalue (kernel:756-816).  The reference forms V sqrt(s) V^T and inverts it with
// LAPACK; here the inverse is applied in the eigenbasis, X = V s^-1/2 V^T, which is the same matrix.
inline int orthogonaliser_device(Workspace &w, int n, const double *dS, double *dX, double *dSinv, double *smallest, double *scratch,
                                 std::string &msg)
{
    const int nn = n * n, g = (nn + 255) / 256;
    double *W = scratch, *Vs = scratch + nn, *vals = scratch + 2 * (size_t)nn, *e = vals + n;
    hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, dS, W, n);
    int rc = eigh(w, n, W, vals, e, msg);
    if (rc) return rc;
    std::vector<double> hv(n);
    TFS_HIP(tfs_memcpy(hv.data(), vals, n * sizeof(double), hipMemcpyDeviceToHost));
    double mn = hv[0];
    for (double v : hv) mn = std::min(mn, v);
    if (smallest) *smallest = mn;
    if (mn < 0) { msg = "A negative overlap matrix eigenvalue was found!"; return TF_ELINALG; }
    // W holds V^T in row-major terms (row k = eigenvector k).  X = V f(s) V^T = (W^T diag) W
    // scale rows of W: Vs[k][i] = W[k][i] * f(s_k)  == scale "columns" of W^T
    for (int mode = 0; mode < 2; ++mode) {
        double *out = mode == 0 ? dX : dSinv;
        if (!out) continue;
        // Vs = diag(f(s)) W  -> use k_scale_cols on the transposed view: element e=(k,i) scaled by s[k]: need row scaling
        // row scaling = column scaling of the transpose; do it with a gemm-free kernel on W^T:
        // form Wt = W^T, scale columns, then X = Wt_scaled * W
        double one = 1.0, zero = 0.0;
        TFS_BLAS(rocblas_dgeam(w.blas, rocblas_operation_transpose, rocblas_operation_none, n, n, &one, W, n, &zero, W, n, Vs, n));
        hipLaunchKernelGGL(k_scale_cols, dim3(g), dim3(256), 0, TFS_ST, Vs, vals, mode, Vs, n);
        TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, Vs, W, 0.0, out));
    }
    return TF_OK;
}

inline int orthogonaliser(Workspace &w, int n, const double *S, double *X, double *S_inv, double *smallest, std::string &msg)
{
    int rc = ensure(w, n, 6, msg);
    if (rc) return rc;
    const size_t nn = (size_t)n * n;
    double *dS = w.pool, *dX = dS + nn, *dSi = dX + nn, *scr = dSi + nn;
    TFS_HIP(tfs_memcpy(dS, S, nn * sizeof(double), hipMemcpyHostToDevice));
    rc = orthogonaliser_device(w, n, dS, dX, S_inv ? dSi : nullptr, smallest, scr, msg);
    if (rc) return rc;
    TFS_HIP(tfs_memcpy(X, dX, nn * sizeof(double), hipMemcpyDeviceToHost));
    if (S_inv) TFS_HIP(tfs_memcpy(S_inv, dSi, nn * sizeof(double), hipMemcpyDeviceToHost));
    return TF_OK;
}

// eps, C = eigh(sym(X^T F X)), C = X C'   (scf:222-250), host buffers
inline int diagonalise(Workspace &w, int n, const double *F, const double *X, double *eps, double *C, std::string &msg)
{
    int rc = ensure(w, n, 6, msg);
    if (rc) return rc;
    const size_t nn = (size_t)n * n;
    const int g = (int)((nn + 255) / 256);
    double *dF = w.pool, *dX = dF + nn, *t1 = dX + nn, *t2 = t1 + nn, *dW = t2 + nn, *vals = dW + nn, *e = vals + n;
    TFS_HIP(tfs_memcpy(dF, F, nn * sizeof(double), hipMemcpyHostToDevice));
    TFS_HIP(tfs_memcpy(dX, X, nn * sizeof(double), hipMemcpyHostToDevice));
    TFS_BLAS(gemm_rm(w.blas, true, false, n, 1.0, dX, dF, 0.0, t1));
    TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dX, 0.0, t2));
    hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t2, dW, n);
    rc = eigh(w, n, dW, vals, e, msg);
    if (rc) return rc;
    TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, dX, dW, 0.0, t1));
    TFS_HIP(tfs_memcpy(eps, vals, n * sizeof(double), hipMemcpyDeviceToHost));
    TFS_HIP(tfs_memcpy(C, t1, nn * sizeof(double), hipMemcpyDeviceToHost));
    return TF_OK;
}

inline double *base_ifail(Workspace &w, int n) { return w.pool + 5 * (size_t)n * n; }

// Instrumentation: seconds per symmetric eigensolve of a random n x n matrix, variant 0 = dsyevd, 1 = dsyev, 2 = dsyevj.
inline int eigh_probe(Workspace &w, int n, int variant, int reps, double *seconds, std::string &msg)
{
    int rc = ensure(w, n, 6, msg);
    if (rc) return rc;
    const size_t nn = (size_t)n * n;
    std::vector<double> h(nn);
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            double v = (double)(st % 2000001ULL) / 1e6 - 1.0;
            h[(size_t)i * n + j] = h[(size_t)j * n + i] = v;
        }
    double *A0 = w.pool, *A = A0 + nn, *V = A + nn, *vals = V + nn, *e = vals + n;
    TFS_HIP(tfs_memcpy(A0, h.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    double tot = 0.0;
    for (int r = 0; r < reps + 1; ++r) {
        TFS_HIP(tfs_memcpy(A, A0, nn * sizeof(double), hipMemcpyDeviceToDevice));
        TFS_HIP(tfs_sync());
        auto t0 = std::chrono::steady_clock::now();
        if (variant == 0) TFS_BLAS(rocsolver_dsyevd(w.blas, rocblas_evect_original, rocblas_fill_upper, n, A, n, vals, e, w.d_info));
        else if (variant == 3) { int rc3 = eigh(w, n, A, vals, e, msg); if (rc3) return rc3; }
        else if (variant == 4 || variant == 5) {
            // partial spectrum: lowest 18 eigenpairs (the occupied orbitals of an Ar2-like SCF)
            rocblas_int *nev = w.d_info + 0;
            const int k = n < 18 ? n : 18;
            if (variant == 4)
                TFS_BLAS(rocsolver_dsyevdx(w.blas, rocblas_evect_original, rocblas_erange_index, rocblas_fill_upper, n, A, n, 0.0, 0.0, 1, k, nev,
                                           vals, V, n, (rocblas_int *)(w.d_scal + 40)));
            else
                TFS_BLAS(rocsolver_dsyevx(w.blas, rocblas_evect_original, rocblas_erange_index, rocblas_fill_upper, n, A, n, 0.0, 0.0, 1, k, 0.0, nev,
                                          vals, V, n, (rocblas_int *)(base_ifail(w, n)), (rocblas_int *)(w.d_scal + 40)));
        }
        else if (variant == 8 || variant == 9) {                  // a batch of four n x n problems in one call (the symmetry blocks of a diatomic)
            // pool: A0 | A | V | vals... : four matrices need 4 nn doubles from A on (ensure() gave 6 nn + vectors): copy A into 4 slots
            if (n > 256) { msg = "probe variant 8: n <= 256"; return TF_EINVAL; }
            double *B = w.pool + nn;                              // 4 matrices of n x n behind A0 (A, V and the tail of the pool)
            for (int q = 1; q < 4; ++q) TFS_HIP(tfs_memcpy(B + q * nn, A0, nn * sizeof(double), hipMemcpyDeviceToDevice));
            double *vb = w.pool + 5 * nn, *eb = vb + 4 * n;       // needs 5 nn + 8 n doubles <= 6 nn + ... for n >= 8
            if (variant == 8)
                TFS_BLAS(rocsolver_dsyevd_strided_batched(w.blas, rocblas_evect_original, rocblas_fill_upper, n, B, n, (rocblas_stride)nn, vb, n, eb, n, w.d_info, 4));
            else
                TFS_BLAS(rocsolver_dsyevdj_strided_batched(w.blas, rocblas_evect_original, rocblas_fill_upper, n, B, n, (rocblas_stride)nn, vb, n, w.d_info, 4));
        }
        else if (variant == 6) TFS_BLAS(rocsolver_dsyevdj(w.blas, rocblas_evect_original, rocblas_fill_upper, n, A, n, vals, w.d_info));
        else if (variant == 7) {                                  // four n x n GEMMs: the cost of one eigenvector refinement step
            for (int q = 0; q < 4; ++q) TFS_BLAS(gemm_rm(w.blas, q & 1, false, n, 1.0, A0, A, 0.0, V));
        }
        else if (variant == 1) TFS_BLAS(rocsolver_dsyev(w.blas, rocblas_evect_original, rocblas_fill_upper, n, A, n, vals, e, w.d_info));
        else {
            double *resid = w.d_scal + 32;
            rocblas_int *nsweeps = w.d_info;
            TFS_BLAS(rocsolver_dsyevj(w.blas, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper, n, A, n, 1e-14, resid, 100,
                                      nsweeps, vals, w.d_info));
        }
        TFS_HIP(tfs_sync());
        if (r > 0) tot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    *seconds = tot / reps;
    return TF_OK;
}

using JKFn = std::function<int(const double *, double *, double *, hipStream_t)>;
// Kohn-Sham hook: V_XC (device [N,N]) and {n_elec, E_X, E_C} for the device density (tf_dft.hip.h); empty for Hartree-Fock
using XCFn = std::function<int(const double *, double *, double *)>;

inline int run_rhf(Workspace &w, int n, const tf_scf_opts &o, const double *S, const double *T, const double *V, const double *Fext,
                   const double *X, const double *P0, double E0, int n_occ, double V_NN, const JKFn &jk, int world,
                   tf_scf_result &out, std::string &msg, const XCFn &xc = XCFn())
{
    // world > 1: the J/K hook completes the partial sums of this rank's tensor rows with the communicator (tf_comm_init) or the registered all-reduce (tf_set_allreduce);
    // every rank then runs the O(N^3) steps redundantly on identical data
    (void)world;
    if (o.max_diis > TF_MAX_DIIS) { msg = "tf_scf_rhf: at most 64 DIIS matrices are held by the native cycle"; return TF_EINVAL; }
    const int max_diis = std::max(1, (int)o.max_diis);
    const int n_mats = 22 + 2 * max_diis;
    int rc = ensure(w, n, n_mats, msg);
    if (rc) return rc;
    TFS_BLAS(rocblas_set_stream(w.blas, TFS_ST));
    const size_t nn = (size_t)n * n;
    const int g = (int)((nn + 255) / 256);
    auto t_wall = std::chrono::steady_clock::now();
    double *base = w.pool;
    auto mat = [&](int k) { return base + (size_t)k * nn; };
    double *dS = mat(0), *dH = mat(1), *dX = mat(2), *dP = mat(3), *dPold = mat(4), *dPbd = mat(5), *dPvold = mat(6), *dPoldbd = mat(7);
    double *dF = mat(8), *dJ = mat(9), *dK = mat(10), *dT = mat(11), *dV = mat(12), *dFx = mat(13), *t1 = mat(14), *t2 = mat(15);
    double *dC = mat(16), *dW = mat(17), *dPn = mat(18), *scr = mat(19), *dVxc = mat(20), *dCsave = mat(21);
    double *hist = mat(22);
    double *vals = base + (size_t)n_mats * nn, *ework = vals + n, *vals_save = ework + n;
    // logical history entry k lives in physical slot slot[k]: trimming the oldest entry renumbers, nothing is copied
    std::vector<int> slot(max_diis);
    for (int k = 0; k < max_diis; ++k) slot[k] = k;
    auto histF = [&](int k) { return hist + (size_t)(2 * slot[k]) * nn; };
    auto histE = [&](int k) { return hist + (size_t)(2 * slot[k] + 1) * nn; };

    TFS_HIP(tfs_memcpy(dS, S, nn * sizeof(double), hipMemcpyHostToDevice));
    TFS_HIP(tfs_memcpy(dT, T, nn * sizeof(double), hipMemcpyHostToDevice));
    TFS_HIP(tfs_memcpy(dV, V, nn * sizeof(double), hipMemcpyHostToDevice));
    if (Fext) TFS_HIP(tfs_memcpy(dFx, Fext, nn * sizeof(double), hipMemcpyHostToDevice));
    else TFS_HIP(hipMemsetAsync(dFx, 0, nn * sizeof(double), TFS_ST));
    TFS_HIP(tfs_memcpy(dP, P0, nn * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, 1.0, dT, 1.0, dV, dH, (int)nn);      // H = T + V (+ field)
    hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, 1.0, dH, 1.0, dFx, dH, (int)nn);
    if (X) TFS_HIP(tfs_memcpy(dX, X, nn * sizeof(double), hipMemcpyHostToDevice));
    else {
        double sm = 0;
        rc = orthogonaliser_device(w, n, dS, dX, nullptr, &sm, scr - 0 /*uses scr..*/, msg);
        if (rc) return rc;
    }
    TFS_HIP(hipMemsetAsync(dPold, 0, nn * sizeof(double), TFS_ST));
    TFS_HIP(hipMemsetAsync(dPbd, 0, nn * sizeof(double), TFS_ST));
    TFS_HIP(hipMemsetAsync(dPvold, 0, nn * sizeof(double), TFS_ST));
    TFS_HIP(hipMemsetAsync(dPoldbd, 0, nn * sizeof(double), TFS_ST));

    TFS_BLAS(rocblas_set_pointer_mode(w.blas, rocblas_pointer_mode_host));
    // device-time spans without synchronising inside the cycle: events from a pool, read after the last iteration
    size_t tev_used = 0;
    std::vector<std::pair<size_t, int>> spans;                    // (first event index, 0 = Fock build, 1 = eigensolver)
    auto span_begin = [&](int kind) -> int {
        while (w.tev.size() < tev_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            w.tev.push_back(e);
        }
        spans.emplace_back(tev_used, kind);
        (void)hipEventRecord(w.tev[tev_used], TFS_ST);
        tev_used += 2;
        return (int)tev_used - 1;
    };
    auto span_end = [&](int idx) { if (idx >= 0) (void)hipEventRecord(w.tev[idx], TFS_ST); };

    // diagonalise F (AO) -> eps, C ; P = 2 C_occ C_occ^T symmetrised      (scf:222-250, 183-211)
    // After the first solve of a cycle the density comes from refined eigenvectors: GEMM-based for n > 64 (where the eigensolver
    // would be rocsolver_dsyevd), one LDS-resident launch for n <= 64 (where it would be the Jacobi kernel).
    static const bool no_refine = getenv("TF_EIGH") != nullptr;
    const bool refining = !no_refine && n >= 2 && n_occ > 0 && n_occ < n;
    bool orbitals_current = false, orbitals_final = false;
    w.ref_n = 0; w.blk_ref_n = 0;
    static const bool no_fused = getenv("TF_REFINE_UNFUSED") != nullptr;
    auto diag_density = [&](const double *Fao, double *Pout) -> int {
        if (refining && !no_fused && n <= TFR_NMAX && w.ref_n == n && !getenv("TF_REFINE_CHECK")) {
            const int te = span_begin(1);
            std::string rmsg;
            const int rr = ref_density_lds(w, n, n_occ, Fao, dX, Pout, 2.0, rmsg);
            span_end(te);
            if (rr == TF_OK) { orbitals_current = false; return TF_OK; }
            if (rr != TF_ELINALG) { msg = rmsg; return rr; }
        }
        TFS_BLAS(gemm_rm(w.blas, true, false, n, 1.0, dX, Fao, 0.0, t1));     // X^T F
        TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dX, 0.0, t2));     // (X^T F) X
        hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t2, dW, n);
        const int te = span_begin(1);
        orbitals_current = false;
        if (refining && w.ref_n == n) {
            double *Xocc = nullptr;
            std::string rmsg;
            const int rr = (n <= TFR_NMAX) ? ref_refine_lds(w, n, n_occ, dW, &Xocc, rmsg)
                         : (w.blk_ref_n == n)  ? ref_refine_blocks(w, n, dW, &Xocc, rmsg) : ref_refine(w, n, n_occ, dW, &Xocc, rmsg);
            if (rr == TF_OK) {
                const double two = 2.0, zero = 0.0;
                TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, Xocc, dX, 0.0, t1));    // rows: occupied orbitals in the AO basis (others 0)
                TFS_BLAS(rocblas_dgemm(w.blas, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &two, t1, n, t1, n, &zero, t2, n));
                hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t2, Pout, n);
                span_end(te);
                if (getenv("TF_REFINE_CHECK")) {                 // debugging aid: the same density from a real eigensolve
                    std::vector<double> hp(nn), hq(nn);
                    TFS_HIP(tfs_memcpy(hp.data(), Pout, nn * sizeof(double), hipMemcpyDeviceToHost));
                    int r2 = eigh(w, n, dW, vals, ework, msg);
                    if (r2) return r2;
                    TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, dX, dW, 0.0, dC));
                    TFS_BLAS(rocblas_dgemm(w.blas, rocblas_operation_transpose, rocblas_operation_none, n, n, n_occ, &two, dC, n, dC, n, &zero, t1, n));
                    hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t1, t2, n);
                    TFS_HIP(tfs_memcpy(hq.data(), t2, nn * sizeof(double), hipMemcpyDeviceToHost));
                    double dmax = 0.0;
                    for (size_t q = 0; q < nn; ++q) dmax = std::max(dmax, std::fabs(hp[q] - hq[q]));
                    fprintf(stderr, "[tf refine] max |P_refined - P_eigh| = %.3e\n", dmax);
                }
                return TF_OK;
            }
            if (rr != TF_ELINALG) { msg = rmsg; return rr; }
        }
        int r = eigh(w, n, dW, vals, ework, msg);
        if (r) return r;
        if (refining) { r = ref_store(w, n, dW, msg, n_occ); if (r) return r; }
        span_end(te);
        TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, dX, dW, 0.0, dC));      // C = X V   (dW rows = eigenvectors)
        const double two = 2.0, zero = 0.0;
        // col-major view of dC is C^T: P = sum_{k<nocc} C[:,k] C[:,k]^T = M[:nocc,:]^T M[:nocc,:]
        TFS_BLAS(rocblas_dgemm(w.blas, rocblas_operation_transpose, rocblas_operation_none, n, n, n_occ, &two, dC, n, dC, n, &zero, t1, n));
        hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t1, Pout, n);
        orbitals_current = true;
        return TF_OK;
    };

    std::vector<double> B((size_t)max_diis * max_diis, 0.0);
    int n_hist = 0;
    w.warm_ok = true;                 // successive Fock matrices are close: warm-start the Jacobi solver from the last eigenvectors
    w.jac_prev_n = 0; w.sym_prev_n = 0;   // (no warm start across cycles: a cycle's result must not depend on what the context solved before)
    struct WarmGuard { Workspace &w; ~WarmGuard() { w.warm_ok = false; w.jac_prev_n = 0; w.sym_prev_n = 0; } } warm_guard{w};
    double E = E0, E_old = E0, commutator = 1.0;
    double comps[7] = {0, 0, 0, 0, 0, 0, 0};
    out.fock_seconds = 0; out.eig_seconds = 0; out.n_iter = 0; out.converged = 0;
    const int nA = (o.n_atoms >= 2) ? o.n_atom_ao[0] : n;

    for (int step = 1; step <= o.max_iter; ++step) {
        E_old = E;
        // P_very_old = P_old ; P_old = P    (scf:1114-1117).  "P_old_before_damping" is the zero matrix in every
        // iteration of the reference: run_restricted_SCF_cycle does not return P_before_damping (scf:1154), so the
        // outer loop keeps handing in its initial zeros (scf:1359,1373).  dPoldbd therefore stays zero.
        std::swap(dPvold, dPold);            // dPvold <- old P_old ; dPold free to be overwritten
        TFS_HIP(hipMemcpyAsync(dPold, dP, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        // Fock matrix (scf:497-531)
        // Kohn-Sham: exchange-correlation matrix and energy densities from the CURRENT (old) density (scf:1121)
        double xc3[3] = {0.0, 0.0, 0.0};
        if (xc) {
            rc = xc(dP, dVxc, xc3);
            if (rc) { msg = "exchange-correlation evaluation failed"; return rc; }
        }
        const int tf = span_begin(0);
        rc = jk(dP, dJ, dK, TFS_ST);
        if (rc) { msg.clear(); return rc; }                          // (the hook has left its message in the context)
        span_end(tf);
        // push into the history: trim to max_diis first (scf:943-946), so that the slot of the new entry is known
        if (n_hist == max_diis) {
            std::rotate(slot.begin(), slot.begin() + 1, slot.end());     // the oldest entry's slot becomes the newest
            for (int r = 0; r + 1 < n_hist; ++r)
                for (int c = 0; c + 1 < n_hist; ++c) B[r * max_diis + c] = B[(r + 1) * max_diis + c + 1];
            --n_hist;
        }
        static const bool no_fused_fock = getenv("TF_FOCK_UNFUSED") != nullptr;
        hipError_t ferr = hipSuccess;
        if (!no_fused_fock && n <= TFR_NMAX &&
            tfref::launch_fock_diis(n, dH, dJ, dK, o.hfx, xc ? dVxc : nullptr, dP, dS, dX, dF, histF(n_hist), histE(n_hist), TFS_ST, &ferr)) {
            // n <= 64: Fock matrix, DIIS error and the history entry in one launch (tf_refine.hip.h)
        } else {
            if (ferr != hipSuccess) { msg = std::string("Fock / DIIS kernel launch failed: ") + hipGetErrorString(ferr); return TF_ENODEVICE; }
            hipLaunchKernelGGL(k_fock, dim3(g), dim3(256), 0, TFS_ST, dH, dJ, dK, o.hfx, t1, (int)nn);
            if (xc) hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, 1.0, t1, 1.0, dVxc, t1, (int)nn);        // + V_XC, scf:525
            hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t1, dF, n);
            // DIIS error e = X^T (F P S - S P F) X   (scf:906-920)
            TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, dF, dP, 0.0, t1));        // F P
            TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dS, 0.0, t2));        // F P S
            TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, dS, dP, 0.0, t1));        // S P
            TFS_BLAS(gemm_rm(w.blas, false, false, n, -1.0, t1, dF, 1.0, t2));       // F P S - S P F
            TFS_BLAS(gemm_rm(w.blas, true, false, n, 1.0, dX, t2, 0.0, t1));         // X^T e
            TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dX, 0.0, histE(n_hist)));   // (X^T e) X
            TFS_HIP(hipMemcpyAsync(histF(n_hist), dF, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        }
        ++n_hist;
        double ee = 0.0;
        {
            double hd[TF_MAX_DIIS];
            for (int k0 = 0; k0 < n_hist; k0 += 8) {               // eight scalar products per launch
                Ptr8 a;
                const int mk = std::min(8, n_hist - k0);
                for (int k = 0; k < 8; ++k) { a.p[k] = histE(0); a.c[k] = 0.0; }
                for (int k = 0; k < mk; ++k) a.p[k] = histE(k0 + k);
                launch_multi_dot(w, histE(n_hist - 1), a, mk, (int)nn, w.d_scal + 128 + k0);
            }
            TFS_HIP(tfs_memcpy(hd, w.d_scal + 128, n_hist * sizeof(double), hipMemcpyDeviceToHost));
            for (int k = 0; k < n_hist; ++k) {
                // the reference stores the error twice (alpha and beta copies, scf:934), hence the factor 2
                B[(n_hist - 1) * max_diis + k] = B[k * max_diis + (n_hist - 1)] = 2.0 * hd[k];
                if (k == n_hist - 1) ee = hd[k];
            }
        }
        commutator = std::sqrt(ee / (double)nn);                                       // scf:918
        // diagonalise, new density, energy with the NEW P and the OLD J,K   (scf:1133-1141)
        rc = diag_density(dF, dPn);
        if (rc) return rc;
        {   // the five energy terms are read back at the end of the iteration, together with the density changes
            Ptr8 a;
            for (int k = 0; k < 8; ++k) { a.p[k] = dT; a.c[k] = 0.0; }
            a.p[0] = dT; a.p[1] = dV; a.p[2] = dFx; a.p[3] = dJ; a.p[4] = dK;
            launch_multi_dot(w, dPn, a, 5, (int)nn, w.d_scal + 18);
        }
        orbitals_final = orbitals_current;                            // of THIS iteration's Fock matrix (the DIIS solve below reuses dC)
        if (orbitals_final && (out.eps || out.C)) {                  // kept on the device; copied out after the cycle
            TFS_HIP(hipMemcpyAsync(vals_save, vals, n * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
            TFS_HIP(hipMemcpyAsync(dCsave, dC, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        }
        // DIIS extrapolation (scf:991-1059)
        double *Pcur = dPn;
        if (step > 2 && o.use_diis && commutator < 0.3) {
            const int m = n_hist + 1;
            std::vector<double> A((size_t)m * m, 0.0), rhs(m, 0.0), x;
            for (int r = 0; r < n_hist; ++r) {
                for (int c = 0; c < n_hist; ++c) A[r * m + c] = B[r * max_diis + c];
                A[r * m + n_hist] = -1.0; A[n_hist * m + r] = -1.0;
            }
            rhs[n_hist] = -1.0;
            if (small_solve(m, A, rhs, x)) {
                for (int k0 = 0; k0 < n_hist; k0 += 8) {
                    Ptr8 a;
                    const int mk = std::min(8, n_hist - k0);
                    for (int k = 0; k < 8; ++k) { a.p[k] = histF(0); a.c[k] = 0.0; }
                    for (int k = 0; k < mk; ++k) { a.p[k] = histF(k0 + k); a.c[k] = x[k0 + k]; }
                    hipLaunchKernelGGL(k_lincomb, dim3(g), dim3(256), 0, TFS_ST, a, mk, scr, (int)nn, k0 > 0 ? 1 : 0);
                }
                rc = diag_density(scr, dPn);     // overwrites dC/vals: results were copied out above
                if (rc) return rc;
            } else {
                n_hist = 0;                        // "Resetting DIIS", scf:1042-1048
            }
        }
        // P_before_damping = P ; damping (scf:763-868)
        TFS_HIP(hipMemcpyAsync(dPbd, Pcur, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        double damp = 0.0;
        if (o.damping == 2) damp = o.damping_factor;
        else if (o.damping == 1 && commutator > 0.01 && step > 1) {
            double pop[4][2];
            const double *dens[4] = {dPbd, dPold, dPoldbd, dPvold};   // A_n_out, A_n1_in, A_n1_out, A_n2_in
            for (int q = 0; q < 4; ++q) {
                hipLaunchKernelGGL(k_mulliken, dim3(1), dim3(256), 0, TFS_ST, dens[q], dS, n, nA, w.d_scal + 2 * q);
            }
            TFS_HIP(tfs_memcpy(&pop[0][0], w.d_scal, 8 * sizeof(double), hipMemcpyDeviceToHost));
            if (o.n_atoms < 2) { for (int q = 0; q < 4; ++q) pop[q][1] = 0.0; }
            double den[2], alpha[2] = {0.0, 0.0};
            for (int a = 0; a < 2; ++a) den[a] = pop[0][a] - pop[2][a] - pop[1][a] + pop[3][a];
            if (den[0] != 0.0 && den[1] != 0.0)
                for (int a = 0; a < 2; ++a) alpha[a] = (pop[0][a] - pop[2][a]) / den[a];
            if (o.n_atoms >= 2) {
                const double r0 = o.n_atom_ao[0], r1 = o.n_atom_ao[1];
                damp = (alpha[0] * r0 + alpha[1] * r1) / (r0 + r1);
            } else damp = alpha[0] * o.n_atom_ao[0];
            damp = std::max(damp, 0.0);
            damp = (damp < std::min(o.max_damping, 1.0)) ? damp : o.max_damping;
        }
        hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, damp, dPold, 1.0 - damp, dPbd, dP, (int)nn);
        // changes and convergence (scf:261-333)
        launch_delta_norms(w, dP, dPold, (int)nn, w.d_scal + 16);
        double res[7];
        TFS_HIP(tfs_memcpy(res, w.d_scal + 16, 7 * sizeof(double), hipMemcpyDeviceToHost));
        {
            const double eT = res[2], eV = res[3], eF = res[4], eJ = res[5], eK = res[6];
            comps[0] = eT; comps[1] = eV; comps[2] = (1.0 / 2.0) * eJ; comps[3] = -(1.0 / 4.0) * eK * o.hfx + xc3[1]; comps[4] = xc3[2];   // scf:380-394
            comps[5] = eF; comps[6] = 0.0;
            E = comps[0] + comps[1] + comps[2] + comps[3] + comps[4] + comps[5] + comps[6];
        }
        const double dE = E - E_old, maxDP = res[0], rmsDP = std::sqrt(res[1] / (double)nn);
        out.n_iter = step;
        if (out.table) {
            double *row = out.table + (size_t)(step - 1) * 7;
            row[0] = step; row[1] = E + V_NN; row[2] = dE; row[3] = rmsDP; row[4] = maxDP; row[5] = commutator; row[6] = damp;
        }
        const bool conv_now = std::fabs(dE) < o.conv_delta_E && std::fabs(maxDP) < o.conv_max_DP && std::fabs(rmsDP) < o.conv_rms_DP &&
                              std::fabs(commutator) < o.conv_commutator;
        if (w.agree) {                                               // collective control flow on a sharded tensor
            const double vals[3] = {conv_now ? 1.0 : 0.0, (double)n_hist, (double)step};
            rc = w.agree(vals, 3, msg);
            if (rc) return rc;
        }
        if (conv_now) {
            out.converged = 1;
            break;
        }
    }
    if (orbitals_final) {
        if (out.eps) TFS_HIP(tfs_memcpy(out.eps, vals_save, n * sizeof(double), hipMemcpyDeviceToHost));
        if (out.C) TFS_HIP(tfs_memcpy(out.C, dCsave, nn * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (!orbitals_final && (out.eps || out.C) && out.n_iter > 0) {
        // orbitals and orbital energies of the last Fock matrix (what the reference's last diagonalisation leaves, scf:1133)
        TFS_BLAS(gemm_rm(w.blas, true, false, n, 1.0, dX, dF, 0.0, t1));
        TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dX, 0.0, t2));
        hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t2, dW, n);
        const int te = span_begin(1);
        rc = eigh(w, n, dW, vals, ework, msg);
        if (rc) return rc;
        TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, dX, dW, 0.0, dC));
        span_end(te);
        if (out.eps) TFS_HIP(tfs_memcpy(out.eps, vals, n * sizeof(double), hipMemcpyDeviceToHost));
        if (out.C) TFS_HIP(tfs_memcpy(out.C, dC, nn * sizeof(double), hipMemcpyDeviceToHost));
    }
    w.ref_n = 0; w.blk_ref_n = 0;
    TFS_HIP(tfs_sync());
    for (const auto &sp : spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, w.tev[sp.first], w.tev[sp.first + 1]) == hipSuccess) (sp.second == 0 ? out.fock_seconds : out.eig_seconds) += ms * 1e-3;
    }
    out.energy = E + V_NN;
    std::memcpy(out.components, comps, sizeof(comps));
    if (out.P) TFS_HIP(tfs_memcpy(out.P, dP, nn * sizeof(double), hipMemcpyDeviceToHost));
    if (out.F) TFS_HIP(tfs_memcpy(out.F, dF, nn * sizeof(double), hipMemcpyDeviceToHost));
    out.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wall).count();
    if (!out.converged) { msg = "Self-consistent field not converged in " + std::to_string(o.max_iter) + " iterations! Increase maximum iterations or give up."; return TF_ENOTCONV; }
    return TF_OK;
}

// ---- unrestricted cycle: run_unrestricted_SCF_cycle (scf:1165-1281) inside the outer loop (scf:1292-1435) ---------------------
// Both spin densities go through the tensor in one fused pass (jk2); every O(N^3) step on the device; per iteration the host reads
// back a handful of scalars (DIIS scalar products, energy terms, populations, density changes).  Reference behaviour kept: the error
// vector of an iteration is the concatenation of the alpha and beta commutators (scf:1213), the commutator reported is the larger
// of the two (scf:1211), each spin is damped with its own factor (scf:1259-1260) in which "P_very_old" and "P_old_before_damping" are
// zero matrices in every iteration (scf:1281 against scf:1394), and the table shows the larger factor.
using JK2Fn = std::function<int(const double *, const double *, double *, double *, double *, double *, hipStream_t)>;

struct UhfOut {
    double *P[2], *C[2], *eps[2], *F[2];                        // host buffers, each may be nullptr
};

inline int run_uhf(Workspace &w, int n, const tf_scf_opts &o, const double *S, const double *T, const double *V, const double *Fext,
                   const double *X, const double *Pa0, const double *Pb0, double E0, int n_alpha, int n_beta, double V_NN, const JK2Fn &jk2,
                   int world, tf_scf_result &out, const UhfOut &uo, std::string &msg)
{
    (void)world;                                                    // (sharded tensors: see run_rhf)
    if (o.max_diis > TF_MAX_DIIS) { msg = "tf_scf_uhf: at most 64 DIIS matrices are held by the native cycle"; return TF_EINVAL; }
    const int max_diis = std::max(1, (int)o.max_diis);
    const int n_fixed = 30;
    const int n_mats = n_fixed + 4 * max_diis;
    int rc = ensure(w, n, n_mats, msg);
    if (rc) return rc;
    TFS_BLAS(rocblas_set_stream(w.blas, TFS_ST));
    const size_t nn = (size_t)n * n;
    const int g = (int)((nn + 255) / 256);
    auto t_wall = std::chrono::steady_clock::now();
    double *base = w.pool;
    auto mat = [&](int k) { return base + (size_t)k * nn; };
    double *dS = mat(0), *dH = mat(1), *dX = mat(2), *dT = mat(3), *dV = mat(4), *dFx = mat(5), *t1 = mat(6), *t2 = mat(7), *dW = mat(8);
    double *dC = mat(9), *scr = mat(10), *dPt = mat(11), *dPtold = mat(12);
    double *dP[2] = {mat(13), mat(14)}, *dPold[2] = {mat(15), mat(16)}, *dPn[2] = {mat(17), mat(18)}, *dF[2] = {mat(19), mat(20)};
    double *dJ[2] = {mat(21), mat(22)}, *dK[2] = {mat(23), mat(24)}, *dCsave[2] = {mat(25), mat(26)};
    double *dJt = mat(27);                                       // J_alpha + J_beta; mat(28), mat(29): orthogonaliser scratch
    double *hist = mat(n_fixed);
    double *vals = base + (size_t)n_mats * nn, *ework = vals + n, *vals_save[2] = {ework + n, ework + 2 * (size_t)n};
    std::vector<int> slot(max_diis);                             // logical history entry -> physical slot (see run_rhf)
    for (int k = 0; k < max_diis; ++k) slot[k] = k;
    auto histF = [&](int k, int s) { return hist + (size_t)(4 * slot[k] + s) * nn; };
    auto histE = [&](int k, int s) { return hist + (size_t)(4 * slot[k] + 2 + s) * nn; };
    const int n_occ[2] = {n_alpha, n_beta};

    TFS_HIP(tfs_memcpy(dS, S, nn * sizeof(double), hipMemcpyHostToDevice));
    TFS_HIP(tfs_memcpy(dT, T, nn * sizeof(double), hipMemcpyHostToDevice));
    TFS_HIP(tfs_memcpy(dV, V, nn * sizeof(double), hipMemcpyHostToDevice));
    if (Fext) TFS_HIP(tfs_memcpy(dFx, Fext, nn * sizeof(double), hipMemcpyHostToDevice));
    else TFS_HIP(hipMemsetAsync(dFx, 0, nn * sizeof(double), TFS_ST));
    TFS_HIP(tfs_memcpy(dP[0], Pa0, nn * sizeof(double), hipMemcpyHostToDevice));
    TFS_HIP(tfs_memcpy(dP[1], Pb0, nn * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, 1.0, dT, 1.0, dV, dH, (int)nn);
    hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, 1.0, dH, 1.0, dFx, dH, (int)nn);
    hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, 1.0, dP[0], 1.0, dP[1], dPt, (int)nn);
    if (X) TFS_HIP(tfs_memcpy(dX, X, nn * sizeof(double), hipMemcpyHostToDevice));
    else {
        double sm = 0;
        rc = orthogonaliser_device(w, n, dS, dX, nullptr, &sm, mat(28), msg);
        if (rc) return rc;
    }
    TFS_BLAS(rocblas_set_pointer_mode(w.blas, rocblas_pointer_mode_host));
    size_t tev_used = 0;
    std::vector<std::pair<size_t, int>> spans;
    auto span_begin = [&](int kind) -> int {
        while (w.tev.size() < tev_used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            w.tev.push_back(e);
        }
        spans.emplace_back(tev_used, kind);
        (void)hipEventRecord(w.tev[tev_used], TFS_ST);
        tev_used += 2;
        return (int)tev_used - 1;
    };
    auto span_end = [&](int idx) { if (idx >= 0) (void)hipEventRecord(w.tev[idx], TFS_ST); };

    static const bool no_refine = getenv("TF_EIGH") != nullptr;
    // the refinement state of the beta spin lives in the alternate slot and is swapped in around its solves
    struct SpinSlot {
        Workspace &w; bool on;
        SpinSlot(Workspace &ws, bool o_) : w(ws), on(o_) { swap(); }
        ~SpinSlot() { swap(); }
        void swap() { if (on) { std::swap(w.ref_buf, w.ref_alt_buf); std::swap(w.ref_cap, w.ref_alt_cap); std::swap(w.ref_n, w.ref_alt_n); std::swap(w.ref_vcls_n, w.ref_alt_vcls_n);
                                 std::swap(w.blk_X, w.blk_alt_X); std::swap(w.blk_ref_n, w.blk_alt_ref_n); std::swap(w.blk_nocc_off, w.blk_alt_nocc_off); } }
    };
    w.ref_n = 0; w.ref_alt_n = 0; w.ref_vcls_n = 0; w.ref_alt_vcls_n = 0; w.blk_ref_n = 0; w.blk_alt_ref_n = 0;
    w.warm_ok = false; w.jac_prev_n = 0; w.sym_prev_n = 0;          // two alternating spins: no warm start for the (rare) Jacobi solves
    bool orbitals_current[2] = {false, false}, orbitals_final[2] = {false, false};
    // diagonalise F_s (AO) -> P_s = C_occ C_occ^T symmetrised (one electron per orbital, scf:1227-1228)
    auto diag_density = [&](int sp, const double *Fao, double *Pout) -> int {
        orbitals_current[sp] = false;
        const int no = n_occ[sp];
        if (no <= 0) { TFS_HIP(hipMemsetAsync(Pout, 0, nn * sizeof(double), TFS_ST)); return TF_OK; }
        SpinSlot slot(w, sp == 1);
        const bool refining = !no_refine && n >= 2 && no < n;
        static const bool no_fused = getenv("TF_REFINE_UNFUSED") != nullptr;
        if (refining && !no_fused && n <= TFR_NMAX && w.ref_n == n) {
            const int te0 = span_begin(1);
            std::string rmsg;
            const int rr = ref_density_lds(w, n, no, Fao, dX, Pout, 1.0, rmsg);
            span_end(te0);
            if (rr == TF_OK) return TF_OK;
            if (rr != TF_ELINALG) { msg = rmsg; return rr; }
        }
        TFS_BLAS(gemm_rm(w.blas, true, false, n, 1.0, dX, Fao, 0.0, t1));
        TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dX, 0.0, t2));
        hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t2, dW, n);
        const int te = span_begin(1);
        const double one = 1.0, zero = 0.0;
        if (refining && w.ref_n == n) {
            double *Xocc = nullptr;
            std::string rmsg;
            const int rr = (n <= TFR_NMAX) ? ref_refine_lds(w, n, no, dW, &Xocc, rmsg)
                         : (w.blk_ref_n == n)  ? ref_refine_blocks(w, n, dW, &Xocc, rmsg) : ref_refine(w, n, no, dW, &Xocc, rmsg);
            if (rr == TF_OK) {
                TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, Xocc, dX, 0.0, t1));
                TFS_BLAS(rocblas_dgemm(w.blas, rocblas_operation_none, rocblas_operation_transpose, n, n, n, &one, t1, n, t1, n, &zero, t2, n));
                hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t2, Pout, n);
                span_end(te);
                return TF_OK;
            }
            if (rr != TF_ELINALG) { msg = rmsg; return rr; }
        }
        int r = eigh(w, n, dW, vals, ework, msg);
        if (r) return r;
        if (refining) { r = ref_store(w, n, dW, msg, no); if (r) return r; }
        span_end(te);
        TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, dX, dW, 0.0, dC));
        TFS_BLAS(rocblas_dgemm(w.blas, rocblas_operation_transpose, rocblas_operation_none, n, n, no, &one, dC, n, dC, n, &zero, t1, n));
        hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t1, Pout, n);
        orbitals_current[sp] = true;
        return TF_OK;
    };

    std::vector<double> B((size_t)max_diis * max_diis, 0.0);
    int n_hist = 0;
    double E = E0, E_old = E0, commutator = 1.0;
    double comps[7] = {0, 0, 0, 0, 0, 0, 0};
    out.fock_seconds = 0; out.eig_seconds = 0; out.n_iter = 0; out.converged = 0;
    const int nA = (o.n_atoms >= 2) ? o.n_atom_ao[0] : n;

    for (int step = 1; step <= o.max_iter; ++step) {
        E_old = E;
        TFS_HIP(hipMemcpyAsync(dPtold, dPt, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        for (int sp = 0; sp < 2; ++sp) TFS_HIP(hipMemcpyAsync(dPold[sp], dP[sp], nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        // Fock matrices (scf:542-589): F_s = H + J_alpha + J_beta - HFX K_s, symmetrised
        const int tf = span_begin(0);
        rc = jk2(dP[0], dP[1], dJ[0], dJ[1], dK[0], dK[1], TFS_ST);
        if (rc) { msg.clear(); return rc; }
        span_end(tf);
        hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, 1.0, dJ[0], 1.0, dJ[1], dJt, (int)nn);
        if (n_hist == max_diis) {                                   // trim the history to max_diis entries (scf:1216-1219)
            std::rotate(slot.begin(), slot.begin() + 1, slot.end());
            for (int r = 0; r + 1 < n_hist; ++r)
                for (int c = 0; c + 1 < n_hist; ++c) B[r * max_diis + c] = B[(r + 1) * max_diis + c + 1];
            --n_hist;
        }
        for (int sp = 0; sp < 2; ++sp) {
            hipLaunchKernelGGL(k_fock, dim3(g), dim3(256), 0, TFS_ST, dH, dJt, dK[sp], 2.0 * o.hfx, t1, (int)nn);   // k_fock: H + J - hfx/2 K
            hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t1, dF[sp], n);
            // e_s = X^T (F_s P_s S - S P_s F_s) X   (scf:906-920)
            TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, dF[sp], dP[sp], 0.0, t1));
            TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dS, 0.0, t2));
            TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, dS, dP[sp], 0.0, t1));
            TFS_BLAS(gemm_rm(w.blas, false, false, n, -1.0, t1, dF[sp], 1.0, t2));
            TFS_BLAS(gemm_rm(w.blas, true, false, n, 1.0, dX, t2, 0.0, t1));
            TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dX, 0.0, histE(n_hist, sp)));
            TFS_HIP(hipMemcpyAsync(histF(n_hist, sp), dF[sp], nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
        }
        ++n_hist;
        double ee[2] = {0.0, 0.0};
        {
            double hd[2 * TF_MAX_DIIS];
            for (int sp = 0; sp < 2; ++sp)
                for (int k0 = 0; k0 < n_hist; k0 += 8) {            // eight scalar products per launch
                    Ptr8 a;
                    const int mk = std::min(8, n_hist - k0);
                    for (int k = 0; k < 8; ++k) { a.p[k] = histE(0, sp); a.c[k] = 0.0; }
                    for (int k = 0; k < mk; ++k) a.p[k] = histE(k0 + k, sp);
                    launch_multi_dot(w, histE(n_hist - 1, sp), a, mk, (int)nn, w.d_scal + 128 + TF_MAX_DIIS * sp + k0);
                }
            TFS_HIP(tfs_memcpy(hd, w.d_scal + 128, 2 * TF_MAX_DIIS * sizeof(double), hipMemcpyDeviceToHost));
            for (int k = 0; k < n_hist; ++k) B[(n_hist - 1) * max_diis + k] = B[k * max_diis + (n_hist - 1)] = hd[k] + hd[TF_MAX_DIIS + k];
            ee[0] = hd[n_hist - 1]; ee[1] = hd[TF_MAX_DIIS + n_hist - 1];
        }
        const double comm_s[2] = {std::sqrt(ee[0] / (double)nn), std::sqrt(ee[1] / (double)nn)};
        commutator = std::max(comm_s[0], comm_s[1]);                  // scf:1211
        // new densities from F_alpha, F_beta; energy with the NEW densities and the OLD J, K (scf:1224-1232)
        for (int sp = 0; sp < 2; ++sp) {
            rc = diag_density(sp, dF[sp], dPn[sp]);
            if (rc) return rc;
            orbitals_final[sp] = orbitals_current[sp];
            if (orbitals_final[sp]) {
                TFS_HIP(hipMemcpyAsync(vals_save[sp], vals, n * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
                TFS_HIP(hipMemcpyAsync(dCsave[sp], dC, nn * sizeof(double), hipMemcpyDeviceToDevice, TFS_ST));
            }
        }
        for (int sp = 0; sp < 2; ++sp) {     // the energy terms are read back at the end of the iteration, together with the density changes
            Ptr8 a;
            for (int k = 0; k < 8; ++k) { a.p[k] = dT; a.c[k] = 0.0; }
            a.p[0] = dT; a.p[1] = dV; a.p[2] = dFx; a.p[3] = dJt; a.p[4] = dK[sp];
            launch_multi_dot(w, dPn[sp], a, 5, (int)nn, w.d_scal + 18 + 6 * sp);
        }
        // DIIS (scf:1236-1256): one set of coefficients for both spins
        if (step > 2 && o.use_diis && commutator < 0.3) {
            const int m = n_hist + 1;
            std::vector<double> A((size_t)m * m, 0.0), rhs(m, 0.0), x;
            for (int r = 0; r < n_hist; ++r) {
                for (int c = 0; c < n_hist; ++c) A[r * m + c] = B[r * max_diis + c];
                A[r * m + n_hist] = -1.0; A[n_hist * m + r] = -1.0;
            }
            rhs[n_hist] = -1.0;
            if (small_solve(m, A, rhs, x)) {
                for (int sp = 0; sp < 2; ++sp) {
                    for (int k0 = 0; k0 < n_hist; k0 += 8) {
                        Ptr8 a;
                        const int mk = std::min(8, n_hist - k0);
                        for (int k = 0; k < 8; ++k) { a.p[k] = histF(0, sp); a.c[k] = 0.0; }
                        for (int k = 0; k < mk; ++k) { a.p[k] = histF(k0 + k, sp); a.c[k] = x[k0 + k]; }
                        hipLaunchKernelGGL(k_lincomb, dim3(g), dim3(256), 0, TFS_ST, a, mk, scr, (int)nn, k0 > 0 ? 1 : 0);
                    }
                    rc = diag_density(sp, scr, dPn[sp]);
                    if (rc) return rc;
                }
            } else
                n_hist = 0;
        }
        // damping, each spin with its own commutator (scf:1259-1260; calculate_damping_factor scf:763-868 with zero
        // "P_very_old" / "P_old_before_damping")
        double damp[2] = {0.0, 0.0};
        if (o.damping == 2) damp[0] = damp[1] = o.damping_factor;
        else if (o.damping == 1 && step > 1 && (comm_s[0] > 0.01 || comm_s[1] > 0.01)) {
            double pop[4][2];
            const double *dens[4] = {dPn[0], dPold[0], dPn[1], dPold[1]};
            for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(k_mulliken, dim3(1), dim3(256), 0, TFS_ST, dens[q], dS, n, nA, w.d_scal + 2 * q);
            TFS_HIP(tfs_memcpy(&pop[0][0], w.d_scal, 8 * sizeof(double), hipMemcpyDeviceToHost));
            if (o.n_atoms < 2) { for (int q = 0; q < 4; ++q) pop[q][1] = 0.0; }
            for (int sp = 0; sp < 2; ++sp) {
                if (!(comm_s[sp] > 0.01)) continue;
                const double *out_p = pop[2 * sp], *in_p = pop[2 * sp + 1];
                double den[2], alpha[2] = {0.0, 0.0};
                for (int a = 0; a < 2; ++a) den[a] = out_p[a] - in_p[a];
                if (den[0] != 0.0 && den[1] != 0.0)
                    for (int a = 0; a < 2; ++a) alpha[a] = out_p[a] / den[a];
                double f;
                if (o.n_atoms >= 2) {
                    const double r0 = o.n_atom_ao[0], r1 = o.n_atom_ao[1];
                    f = (alpha[0] * r0 + alpha[1] * r1) / (r0 + r1);
                } else f = alpha[0] * o.n_atom_ao[0];
                f = std::max(f, 0.0);
                damp[sp] = (f < std::min(o.max_damping, 1.0)) ? f : o.max_damping;
            }
        }
        for (int sp = 0; sp < 2; ++sp)
            hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, damp[sp], dPold[sp], 1.0 - damp[sp], dPn[sp], dP[sp], (int)nn);
        hipLaunchKernelGGL(k_axpby, dim3(g), dim3(256), 0, TFS_ST, 1.0, dP[0], 1.0, dP[1], dPt, (int)nn);
        launch_delta_norms(w, dPt, dPtold, (int)nn, w.d_scal + 16);
        double res[14];
        TFS_HIP(tfs_memcpy(res, w.d_scal + 16, 14 * sizeof(double), hipMemcpyDeviceToHost));
        {
            const double *hd = res + 2;
            comps[0] = hd[0] + hd[6]; comps[1] = hd[1] + hd[7]; comps[5] = hd[2] + hd[8];
            comps[2] = (1.0 / 2.0) * (hd[3] + hd[9]);                                        // scf:462
            comps[3] = -(1.0 / 2.0) * hd[4] * o.hfx + -(1.0 / 2.0) * hd[10] * o.hfx;         // scf:465-466
            comps[4] = 0.0; comps[6] = 0.0;
            E = comps[0] + comps[1] + comps[2] + comps[3] + comps[4] + comps[5] + comps[6];
        }
        const double dE = E - E_old, maxDP = res[0], rmsDP = std::sqrt(res[1] / (double)nn);
        out.n_iter = step;
        if (out.table) {
            double *row = out.table + (size_t)(step - 1) * 7;
            row[0] = step; row[1] = E + V_NN; row[2] = dE; row[3] = rmsDP; row[4] = maxDP; row[5] = commutator; row[6] = std::max(damp[0], damp[1]);
        }
        const bool conv_now = std::fabs(dE) < o.conv_delta_E && std::fabs(maxDP) < o.conv_max_DP && std::fabs(rmsDP) < o.conv_rms_DP &&
                              std::fabs(commutator) < o.conv_commutator;
        if (w.agree) {                                               // collective control flow on a sharded tensor
            const double vals[3] = {conv_now ? 1.0 : 0.0, (double)n_hist, (double)step};
            rc = w.agree(vals, 3, msg);
            if (rc) return rc;
        }
        if (conv_now) {
            out.converged = 1;
            break;
        }
    }
    // orbitals and orbital energies of the last Fock matrices (what the reference's last diagonalisations leave, scf:1224-1225)
    for (int sp = 0; sp < 2 && out.n_iter > 0; ++sp) {
        if (!uo.eps[sp] && !uo.C[sp]) continue;
        if (orbitals_final[sp]) {
            if (uo.eps[sp]) TFS_HIP(tfs_memcpy(uo.eps[sp], vals_save[sp], n * sizeof(double), hipMemcpyDeviceToHost));
            if (uo.C[sp]) TFS_HIP(tfs_memcpy(uo.C[sp], dCsave[sp], nn * sizeof(double), hipMemcpyDeviceToHost));
            continue;
        }
        TFS_BLAS(gemm_rm(w.blas, true, false, n, 1.0, dX, dF[sp], 0.0, t1));
        TFS_BLAS(gemm_rm(w.blas, false, false, n, 1.0, t1, dX, 0.0, t2));
        hipLaunchKernelGGL(k_symmetrise, dim3(g), dim3(256), 0, TFS_ST, t2, dW, n);
        const int te = span_begin(1);
        rc = eigh(w, n, dW, vals, ework, msg);
        if (rc) return rc;
        TFS_BLAS(gemm_rm(w.blas, false, true, n, 1.0, dX, dW, 0.0, dC));
        span_end(te);
        if (uo.eps[sp]) TFS_HIP(tfs_memcpy(uo.eps[sp], vals, n * sizeof(double), hipMemcpyDeviceToHost));
        if (uo.C[sp]) TFS_HIP(tfs_memcpy(uo.C[sp], dC, nn * sizeof(double), hipMemcpyDeviceToHost));
    }
    w.ref_n = 0; w.ref_alt_n = 0; w.ref_vcls_n = 0; w.ref_alt_vcls_n = 0; w.blk_ref_n = 0; w.blk_alt_ref_n = 0;
    TFS_HIP(tfs_sync());
    for (const auto &sp : spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, w.tev[sp.first], w.tev[sp.first + 1]) == hipSuccess) (sp.second == 0 ? out.fock_seconds : out.eig_seconds) += ms * 1e-3;
    }
    out.energy = E + V_NN;
    std::memcpy(out.components, comps, sizeof(comps));
    for (int sp = 0; sp < 2; ++sp) {
        if (uo.P[sp]) TFS_HIP(tfs_memcpy(uo.P[sp], dP[sp], nn * sizeof(double), hipMemcpyDeviceToHost));
        if (uo.F[sp]) TFS_HIP(tfs_memcpy(uo.F[sp], dF[sp], nn * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (out.P) TFS_HIP(tfs_memcpy(out.P, dPt, nn * sizeof(double), hipMemcpyDeviceToHost));
    out.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wall).count();
    if (!out.converged) { msg = "Self-consistent field not converged in " + std::to_string(o.max_iter) + " iterations! Increase maximum iterations or give up."; return TF_ENOTCONV; }
    return TF_OK;
}

}  // namespace tfscf
