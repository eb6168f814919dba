// tf_mp2.hip.h -- AO->MO integral transformation on the HBM-resident tensor and the RMP2 energy (SURVEY.md section 8f, rank 1;
// BASELINE config 5).  This is the one place of the path that is GEMM-shaped: every quarter transformation is a (batched) f64
// GEMM through rocBLAS (MFMA f64 on gfx950).
// Reference: transform_ERI_AO_to_MO tuna_ci.py:204-255 (four einsums over the dense N^4 tensor),
//            build_doubles_epsilons_tensor tuna_ci.py:304-334, run_restricted_MP2 tuna_mp.py:834-906 (energy part).
// The stored tensor keeps rows (mu >= nu), so the ket half-transformation
//     Q[mu nu][r s] = sum_{lambda sigma} C3[lambda r] (mu nu|lambda sigma) C4[sigma s]
// is two strided-batched GEMMs with one stored row per batch entry, and the bra half is two more GEMMs on the unpacked Q.
// Rows layout: a row is the full [lambda][sigma] matrix.  Packed layout: a row holds the pairs (lambda sigma) <= (mu nu) only; it is
// expanded to a symmetric matrix of that part, and the other half of the tensor comes from the transposed result (see transform()).
// Each stored value is read once and only from its own row, so a sharded tensor transforms rank by rank (sum = all-reduce).
#pragma once
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>

#include <algorithm>
#include <string>

#include "../../include/tunafock.h"
#include "tf_jkpacked.hip.h"

namespace tfmp2 {

#define TFM_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (call);                                                                         \
        if (_e != hipSuccess) { msg = std::string(#call) + " failed: " + hipGetErrorString(_e); rc = (_e == hipErrorOutOfMemory ? TF_ENOMEM : TF_ENODEVICE); goto done; } \
    } while (0)
#define TFM_BLAS(call)                                                                                  \
    do {                                                                                                \
        rocblas_status _s = (call);                                                                     \
        if (_s != rocblas_status_success) { msg = std::string(#call) + " failed (rocBLAS status " + std::to_string((int)_s) + ")"; rc = TF_ELINALG; goto done; } \
    } while (0)

// Qfull[mu][nu][x] = Q[row(max,min)][x]; the row table of the packed layout is keyed by internal AO indices (ao: class | loc << 2
// of every original AO, nullptr for the rows layout)
__global__ void unpack_rows_kernel(const double *__restrict__ Q, const int *__restrict__ rowmap, int N, long long width,
                                   double *__restrict__ Qfull, BLayout L, int packed, const int *__restrict__ row_pos /* packed: where Q keeps row r */)
{
    const long long total = (long long)N * N * width;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long mn = e / width;
        const long long x = e - mn * width;
        int mu = (int)(mn / N), nu = (int)(mn - (long long)mu * N);
        if (packed) { mu = ao_sigma(L, L.ao[mu]); nu = ao_sigma(L, L.ao[nu]); }
        const int hi = max(mu, nu), lo = min(mu, nu);
        int r = rowmap[hi * (hi + 1) / 2 + lo];
        if (r >= 0 && packed) r = row_pos[r];
        Qfull[e] = (r >= 0) ? Q[(long long)r * width + x] : 0.0;
    }
}

// RMP2 energy from g[i][a][j][b] = (ia|jb):  E_OS = sum g^2 / D,  E_SS = sum g (g - g[i][b][j][a]) / D   (tuna_mp.py:882-890)
__global__ void mp2_energy_kernel(const double *__restrict__ g, const double *__restrict__ eps, int n_frozen, int o, int v, int n_occ_total,
                                  double *__restrict__ partial /* [gridDim.x][2] */)
{
    __shared__ double s_os[256], s_ss[256];
    const long long total = (long long)o * v * o * v;
    double os = 0.0, ss = 0.0;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        long long r = e;
        const int b = (int)(r % v); r /= v;
        const int j = (int)(r % o); r /= o;
        const int a = (int)(r % v);
        const int i = (int)(r / v);
        const double gij = g[e];
        const double gx = g[(((long long)i * v + b) * o + j) * v + a];
        const double D = eps[n_frozen + i] + eps[n_frozen + j] - eps[n_occ_total + a] - eps[n_occ_total + b];
        os += gij * gij / D;
        ss += gij * (gij - gx) / D;
    }
    s_os[threadIdx.x] = os; s_ss[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { s_os[threadIdx.x] += s_os[threadIdx.x + s]; s_ss[threadIdx.x] += s_ss[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = s_os[0]; partial[2 * blockIdx.x + 1] = s_ss[0]; }
}

// Ci[x][:] = C[origI[x]][:]: coefficient rows in the internal (class-sorted) AO order of the packed layout
__global__ void permute_rows_kernel(const double *__restrict__ C, const int *__restrict__ origI, int N, int n, double *__restrict__ Ci)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * n) return;
    const int x = e / n, p = e - x * n;
    Ci[e] = C[(size_t)origI[x] * n + p];
}

// host view of the packed layout for the transformation: class sizes / starts (internal order), pair spaces, rows listed by class
struct PackedRows {
    int csize[4], cstart[4];
    long long NP[4], class_row_off[5];
    const int *d_class_rows, *d_row_pos;
};

// out[p][q][r][s] = sum C1[mu p] C2[nu q] C3[la r] C4[si s] (mu nu|la si); C_k are [N, n_k] row-major DEVICE matrices;
// d_out [n1,n2,n3,n4] on the device.  Rows of the stored tensor are processed in slabs to bound the scratch.
// Packed layout (d_rowoff != nullptr): the STORED part of every row of a slab is materialised as a symmetric [N][ld] matrix
// (unpack_own_rows_kernel), so d_out is the transform of L, the tensor restricted to (kl) <= (ij) with its diagonal halved; the caller
// completes it with the transform of L^T (mo_transform_device: G(C1 C2 C3 C4)[pq][rs] + G(C3 C4 C1 C2)[rs][pq]).  Either way the
// result is linear in the rows a rank owns: the sum over ranks is the transformed tensor.
inline int transform(rocblas_handle blas, const double *d_eri, const int *d_rowmap, const long long *d_rowoff, const int *d_rowsec,
                     const BLayout &BL, const PackedRows &PR, const int2 *d_row_ij, long long n_rows, int N, int ld, const double *dC1, int n1, const double *dC2, int n2, const double *dC3, int n3,
                     const double *dC4, int n4, double *d_out, double *gemm_seconds, std::string &msg)
{
    int rc = TF_OK;
    const long long n34 = (long long)n3 * n4;
    const long long row_len = (long long)N * ld;
    double *dR = nullptr, *dQ = nullptr, *dQfull = nullptr, *dW = nullptr, *dM = nullptr, *dC3i = nullptr, *dC4i = nullptr;
    const bool packed = d_rowoff != nullptr;
    const double one = 1.0, zero = 0.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    RowBlocks RBc[4];
    long long max_rs = 1;
    // slab of rows for the first quarter transformation: R = C3^T M needs n3*N doubles per row
    long long slab = std::max<long long>(1, std::min<long long>(n_rows, (long long)((2048LL << 20) / ((long long)n3 * N * sizeof(double)))));
    if (n34 * N > 0x7fffffffLL || n34 > 0x7fffffffLL) { msg = "AO->MO transformation: dimension overflow"; return TF_EINVAL; }
    TFM_HIP(hipEventCreate(&e0));
    TFM_HIP(hipEventCreate(&e1));
    // packed: the largest class-blocked row (doubles) bounds the slab as well
    if (packed) {
        for (int c = 0; c < 4; ++c) {
            long long o = 0;
            for (int a = 0; a < 4; ++a) { RBc[c].boff[a] = (int)o; RBc[c].ldb[a] = std::max(1, PR.csize[a ^ c]); o += (long long)PR.csize[a] * RBc[c].ldb[a]; }
            RBc[c].rstride = (o + 1) & ~1LL;
            max_rs = std::max(max_rs, RBc[c].rstride);
        }
        slab = std::max<long long>(1, std::min<long long>(std::min<long long>(slab, 65535), (long long)((2048LL << 20) / (max_rs * (long long)sizeof(double)))));
        TFM_HIP(hipMalloc((void **)&dM, (size_t)slab * max_rs * sizeof(double)));
        TFM_HIP(hipMalloc((void **)&dC3i, (size_t)N * n3 * sizeof(double)));
        TFM_HIP(hipMalloc((void **)&dC4i, (size_t)N * n4 * sizeof(double)));
        hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((N * n3 + 255) / 256)), dim3(256), 0, 0, dC3, BL.origI, N, n3, dC3i);
        hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((N * n4 + 255) / 256)), dim3(256), 0, 0, dC4, BL.origI, N, n4, dC4i);
    }
    TFM_HIP(hipMalloc((void **)&dR, (size_t)slab * n3 * N * sizeof(double)));
    TFM_HIP(hipMalloc((void **)&dQ, (size_t)std::max<long long>(1, n_rows) * n34 * sizeof(double)));
    TFM_HIP(hipEventRecord(e0, 0));
    if (!packed) {
        for (long long r0 = 0; r0 < n_rows; r0 += slab) {
            const int nb = (int)std::min<long long>(slab, n_rows - r0);
            const double *Mrows = d_eri + r0 * row_len;
            // R[row] (n3 x N, row-major) = C3^T (n3 x N) * M[row] (N x N, ld)
            TFM_BLAS(rocblas_dgemm_strided_batched(blas, rocblas_operation_none, rocblas_operation_transpose, N, n3, N, &one, Mrows, ld, row_len,
                                                   dC3, n3, 0, &zero, dR, N, (rocblas_stride)n3 * N, nb));
            // Q[row] (n3 x n4) = R[row] (n3 x N) * C4 (N x n4)
            TFM_BLAS(rocblas_dgemm_strided_batched(blas, rocblas_operation_none, rocblas_operation_none, n4, n3, N, &one, dC4, n4, 0, dR, N,
                                                   (rocblas_stride)n3 * N, &zero, dQ + r0 * n34, n4, (rocblas_stride)n34, nb));
        }
    } else {
        // class by class: the stored part of a class-c row is nonzero only in the blocks (k of class a) x (l of class a ^ c), a quarter of
        // the N x N matrix; Q is kept in class order (row_pos).  Internal AO order throughout the ket half.
        for (int c = 0; c < 4; ++c) {
            const long long nrc = PR.class_row_off[c + 1] - PR.class_row_off[c];
            const RowBlocks &RB = RBc[c];
            for (long long t0 = 0; t0 < nrc; t0 += slab) {
                const int nb = (int)std::min<long long>(slab, nrc - t0);
                const long long q0 = PR.class_row_off[c] + t0;
                TFM_HIP(hipMemsetAsync(dM, 0, (size_t)nb * RB.rstride * sizeof(double), 0));
                TFM_HIP(hipMemsetAsync(dR, 0, (size_t)nb * n3 * N * sizeof(double), 0));      // (a class without members leaves its columns untouched)
                hipLaunchKernelGGL(unpack_own_rows_blocked_kernel, dim3((unsigned)((PR.NP[c] + 255) / 256), (unsigned)nb), dim3(256), 0, 0, d_eri,
                                   d_rowoff, d_rowsec, BL, d_row_ij, PR.d_class_rows + q0, c, RB, dM);
                for (int b = 0; b < 4; ++b) {
                    const int a = b ^ c;
                    if (PR.csize[b] == 0 || PR.csize[a] == 0) continue;
                    // R[row][p][cstart[b] + l] = sum_{k in a} C3i[cstart[a] + k][p] M_a[k][l]   (column-major: (|b| x |a|) (|a| x n3))
                    TFM_BLAS(rocblas_dgemm_strided_batched(blas, rocblas_operation_none, rocblas_operation_transpose, PR.csize[b], n3, PR.csize[a], &one,
                                                           dM + RB.boff[a], RB.ldb[a], (rocblas_stride)RB.rstride, dC3i + (size_t)PR.cstart[a] * n3, n3, 0,
                                                           &zero, dR + PR.cstart[b], N, (rocblas_stride)n3 * N, nb));
                }
                TFM_BLAS(rocblas_dgemm_strided_batched(blas, rocblas_operation_none, rocblas_operation_none, n4, n3, N, &one, dC4i, n4, 0, dR, N,
                                                       (rocblas_stride)n3 * N, &zero, dQ + q0 * n34, n4, (rocblas_stride)n34, nb));
            }
        }
    }
    (void)hipFree(dR); dR = nullptr;
    if (dM) { (void)hipFree(dM); dM = nullptr; }
    TFM_HIP(hipMalloc((void **)&dQfull, (size_t)N * N * n34 * sizeof(double)));
    {
        const long long tot = (long long)N * N * n34;
        hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)std::min<long long>((tot + 255) / 256, 1 << 20)), dim3(256), 0, 0, dQ, d_rowmap, N,
                           n34, dQfull, BL, packed ? 1 : 0, PR.d_row_pos);
    }
    TFM_HIP(hipDeviceSynchronize());
    (void)hipFree(dQ); dQ = nullptr;
    TFM_HIP(hipMalloc((void **)&dW, (size_t)n1 * N * n34 * sizeof(double)));
    // W (n1 x N*n34) = C1^T (n1 x N) * Qfull (N x N*n34)
    TFM_BLAS(rocblas_dgemm(blas, rocblas_operation_none, rocblas_operation_transpose, (rocblas_int)(N * n34), n1, N, &one, dQfull,
                           (rocblas_int)(N * n34), dC1, n1, &zero, dW, (rocblas_int)(N * n34)));
    // out[p] (n2 x n34) = C2^T (n2 x N) * W[p] (N x n34)
    TFM_BLAS(rocblas_dgemm_strided_batched(blas, rocblas_operation_none, rocblas_operation_transpose, (rocblas_int)n34, n2, N, &one, dW,
                                           (rocblas_int)n34, (rocblas_stride)N * n34, dC2, n2, 0, &zero, d_out, (rocblas_int)n34,
                                           (rocblas_stride)n2 * n34, n1));
    TFM_HIP(hipEventRecord(e1, 0));
    TFM_HIP(hipEventSynchronize(e1));
    if (gemm_seconds) { float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1); *gemm_seconds = ms * 1e-3; }
done:
    if (dR) (void)hipFree(dR);
    if (dM) (void)hipFree(dM);
    if (dC3i) (void)hipFree(dC3i);
    if (dC4i) (void)hipFree(dC4i);
    if (dQ) (void)hipFree(dQ);
    if (dQfull) (void)hipFree(dQfull);
    if (dW) (void)hipFree(dW);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

}  // namespace tfmp2
