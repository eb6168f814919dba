// tf_jacobi.hip.h -- symmetric eigensolver for the small matrices of the SCF cycle (n <= 140): parallel cyclic Jacobi,
// one workgroup, matrix (and, when it fits, the eigenvectors) resident in LDS.
// Why: at N = 60..120 the SCF iteration is dominated by the eigensolver, not by the Fock build (J/K of N2/cc-pVTZ takes
// 30 us, rocSOLVER dsyevd 1.2 ms because it is a chain of ~100 tiny launches); a single-launch in-LDS Jacobi removes that
// launch latency.  Larger matrices go to rocsolver_dsyevd.  Reference role: np.linalg.eigh in diagonalise_Fock_matrix
// (scf:244) and calculate_orthogonalisation_matrix (kernel:784).
// Output convention (same as tfscf::eigh): eigenvalues ascending in vals; W row k = eigenvector k.
#pragma once
#include <hip/hip_runtime.h>

namespace tfjac {

#ifndef TFJ_THREADS
#define TFJ_THREADS 1024
#endif
#define TFJ_NMAX 140

// round-robin tournament: pair k of round r among m (even) players
__device__ __forceinline__ void tournament_pair(int m, int r, int k, int &p, int &q)
{
    int a, b;
    if (k == 0) { a = m - 1; b = r; }
    else { a = (r + k) % (m - 1); b = (r - k + (m - 1)) % (m - 1); }
    p = min(a, b); q = max(a, b);
}

// V0 (optional, V_IN_LDS only): rows = eigenvectors of a nearby matrix (the previous SCF iteration).  The sweeps then start
// from A0 = V0 A V0^T, which is already almost diagonal, and converge in 2-3 sweeps instead of ~8.
template <bool V_IN_LDS>
__global__ __launch_bounds__(TFJ_THREADS) void jacobi_eigh_kernel(int n, double *__restrict__ W /* in: A, out: eigenvector rows */,
                                                                  double *__restrict__ vals, double *__restrict__ Vg /* n*n scratch */,
                                                                  const double *__restrict__ V0, double *__restrict__ Vkeep,
                                                                  int max_sweeps, int *__restrict__ info,
                                                                  const int *__restrict__ sizes = nullptr, long long stride = 0, int vstride = 0)
{
    // a batch (the symmetry blocks of a matrix, tf_scf.hip.h: eigh_blocked): workgroup b solves the sizes[b] x sizes[b] matrix at
    // W + b stride (compact: leading dimension sizes[b]); values at vals + b vstride, info[b]
    if (sizes) {
        const long long off = (long long)blockIdx.x * stride;
        n = sizes[blockIdx.x];
        W += off; vals += (size_t)blockIdx.x * vstride;
        if (Vg) Vg += off;
        if (V0) V0 += off;
        if (Vkeep) Vkeep += off;
        if (info) info += blockIdx.x;
        if (n < 1) return;
    }
    extern __shared__ double sm[];
    const int lda = n | 1;
    double *sA = sm;
    double *sV = V_IN_LDS ? sA + (size_t)n * lda : Vg;       // row p = current eigenvector estimate p
    const int ldv = V_IN_LDS ? lda : n;
    double *sC = sm + (V_IN_LDS ? 2 : 1) * (size_t)n * lda;
    double *sS = sC + (TFJ_NMAX / 2 + 1);
    double *sRed = sS + (TFJ_NMAX / 2 + 1);                  // 2 * TFJ_THREADS
    int *sPQ = reinterpret_cast<int *>(sRed + 2 * TFJ_THREADS);   // packed p | q << 16, or -1
    __shared__ int sDone;

    const int tid = threadIdx.x;
    const bool warm = V_IN_LDS && V0 != nullptr;
    for (int e = tid; e < n * n; e += TFJ_THREADS) {
        const int i = e / n, j = e - i * n;
        sA[i * lda + j] = 0.5 * (W[e] + W[(size_t)j * n + i]);
        sV[i * ldv + j] = warm ? V0[e] : ((i == j) ? 1.0 : 0.0);
    }
    const int m = n + (n & 1), half = m / 2;
    int sweeps = 0;
    __syncthreads();
    if (warm) {
        // third LDS matrix (after the int scratch): T = A V^T, then A0 = V T, upper triangle mirrored
        double *sT = reinterpret_cast<double *>(reinterpret_cast<char *>(sPQ) + (((TFJ_NMAX + 2) * sizeof(int) + 15) & ~size_t(15)));
        for (int e = tid; e < n * n; e += TFJ_THREADS) {
            const int i = e / n, k = e - i * n;
            double t = 0.0;
            for (int j = 0; j < n; ++j) t += sA[i * lda + j] * sV[k * ldv + j];
            sT[i * lda + k] = t;
        }
        __syncthreads();
        for (int e = tid; e < n * n; e += TFJ_THREADS) {
            const int k = e / n, l = e - k * n;
            if (l < k) continue;
            double t = 0.0;
            for (int i = 0; i < n; ++i) t += sV[k * ldv + i] * sT[i * lda + l];
            sA[k * lda + l] = t;
            sA[l * lda + k] = t;
        }
        __syncthreads();
    }
    const int lane = tid & 63, wv = tid >> 6;
    for (; sweeps < max_sweeps; ++sweeps) {
        // convergence: off-diagonal weight relative to the whole matrix (wave shuffles, then the 16 wave sums: two barriers)
        double off = 0.0, tot = 0.0;
        for (int e = tid; e < n * n; e += TFJ_THREADS) {
            const int i = e / n, j = e - i * n;
            const double a = sA[i * lda + j];
            tot += a * a;
            if (i != j) off += a * a;
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { off += __shfl_xor(off, d, 64); tot += __shfl_xor(tot, d, 64); }
        if (lane == 0) { sRed[wv] = off; sRed[TFJ_THREADS + wv] = tot; }
        __syncthreads();
        if (tid == 0) {
            double o2 = 0.0, t2 = 0.0;
            for (int u = 0; u < TFJ_THREADS / 64; ++u) { o2 += sRed[u]; t2 += sRed[TFJ_THREADS + u]; }
            sDone = (o2 <= 1e-31 * t2) ? 1 : 0;
        }
        __syncthreads();
        if (sDone) break;
        for (int r = 0; r < m - 1; ++r) {
            if (tid < half) {
                int p, q;
                tournament_pair(m, r, tid, p, q);
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = sA[p * lda + q];
                    if (apq != 0.0) {
                        const double tau = (sA[q * lda + q] - sA[p * lda + p]) / (2.0 * apq);
                        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + t * t);
                        s = t * c;
                    }
                }
                sPQ[tid] = p | (q << 16); sC[tid] = c; sS[tid] = s;          // (q == n: the bye of an odd n -- identity, its row / column does not exist)
            }
            __syncthreads();
            // A <- J^T A J in ONE phase: the pairs of the round are disjoint, so the 2 x 2 block (rows p, q of pair k) x (columns p', q' of
            // pair k') belongs to one thread, which applies the column rotation of k' and the row rotation of k (three barriers and two
            // passes over A per round before: the round is bound by barriers and LDS latency, not arithmetic); rows of V in the same phase
            for (int e = tid; e < half * half + half * n; e += TFJ_THREADS) {
                if (e < half * half) {
                    const int k = e / half, k2 = e - k * half;
                    const int cr = sPQ[k], cc = sPQ[k2];
                    const int p = cr & 0xffff, q = cr >> 16, p2 = cc & 0xffff, q2 = cc >> 16;
                    const bool vq = q < n, vq2 = q2 < n;
                    const double c = sC[k], sn = sS[k], c2 = sC[k2], s2 = sS[k2];
                    const double b00 = sA[p * lda + p2], b01 = vq2 ? sA[p * lda + q2] : 0.0;
                    const double b10 = vq ? sA[q * lda + p2] : 0.0, b11 = (vq && vq2) ? sA[q * lda + q2] : 0.0;
                    const double t00 = c2 * b00 - s2 * b01, t01 = s2 * b00 + c2 * b01;
                    const double t10 = c2 * b10 - s2 * b11, t11 = s2 * b10 + c2 * b11;
                    const bool dg = k == k2 && sn != 0.0;                      // the rotated pair itself: its off-diagonal element is annihilated
                    sA[p * lda + p2] = c * t00 - sn * t10;
                    if (vq2) sA[p * lda + q2] = dg ? 0.0 : c * t01 - sn * t11;
                    if (vq) sA[q * lda + p2] = dg ? 0.0 : sn * t00 + c * t10;
                    if (vq && vq2) sA[q * lda + q2] = sn * t01 + c * t11;
                } else {
                    const int f = e - half * half, k = f / n, j = f - k * n;
                    const int cr = sPQ[k];
                    const int p = cr & 0xffff, q = cr >> 16;
                    const double c = sC[k], sn = sS[k];
                    if (q < n && sn != 0.0) {
                        const double vp = sV[p * ldv + j], vq = sV[q * ldv + j];
                        sV[p * ldv + j] = c * vp - sn * vq;
                        sV[q * ldv + j] = sn * vp + c * vq;
                    }
                }
            }
            __syncthreads();
        }
    }
    // sort ascending (stable rank) and write out
    for (int i = tid; i < n; i += TFJ_THREADS) {
        const double di = sA[i * lda + i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double dj = sA[j * lda + j];
            rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
        }
        sPQ[i] = rank;                       // safe: the pair codes are no longer needed
        vals[rank] = di;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += TFJ_THREADS) {
        const int i = e / n, j = e - i * n;
        const double v = sV[i * ldv + j];
        W[(size_t)sPQ[i] * n + j] = v;
        if (Vkeep) Vkeep[(size_t)sPQ[i] * n + j] = v;
    }
    if (tid == 0 && info) *info = (sweeps >= max_sweeps) ? 1 : 0;
}

inline size_t lds_bytes(int n, bool v_in_lds, bool warm = false)
{
    const size_t lda = (size_t)(n | 1);
    size_t d = (v_in_lds ? 2 : 1) * (size_t)n * lda + 2 * (TFJ_NMAX / 2 + 1) + 2 * TFJ_THREADS;
    size_t b = d * sizeof(double) + (((size_t)(TFJ_NMAX + 2) * sizeof(int) + 15) & ~size_t(15));
    if (warm) b += (size_t)n * lda * sizeof(double);
    return b;
}

// returns false if n is outside the kernel's range (caller falls back to rocSOLVER)
inline bool launch(int n, double *W, double *vals, double *Vscratch, int *info, hipStream_t st, hipError_t *err,
                   const double *V0 = nullptr, double *Vkeep = nullptr)
{
    static bool attr_set = false;
    if (n < 2 || n > TFJ_NMAX) return false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)jacobi_eigh_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        (void)hipFuncSetAttribute((const void *)jacobi_eigh_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    const size_t cap = 160 * 1024 - 256;
    const bool v_in = lds_bytes(n, true) <= cap;
    if (!v_in && lds_bytes(n, false) > cap) return false;
    if (v_in) {
        const bool warm = V0 != nullptr && lds_bytes(n, true, true) <= cap;
        hipLaunchKernelGGL(jacobi_eigh_kernel<true>, dim3(1), dim3(TFJ_THREADS), lds_bytes(n, true, warm), st, n, W, vals, Vscratch,
                           warm ? V0 : nullptr, Vkeep, 40, info);
    } else
        hipLaunchKernelGGL(jacobi_eigh_kernel<false>, dim3(1), dim3(TFJ_THREADS), lds_bytes(n, false), st, n, W, vals, Vscratch,
                           (const double *)nullptr, Vkeep, 40, info);
    *err = hipGetLastError();
    return *err == hipSuccess;
}

// a batch of nb problems of at most mmax <= 64 rows each in one launch (eigenvectors in LDS; warm start per block as above)
inline bool launch_batch(int nb, int mmax, const int *d_sizes, double *W, long long stride, double *vals, int vstride, int *info,
                         hipStream_t st, hipError_t *err, const double *V0 = nullptr, double *Vkeep = nullptr)
{
    if (nb < 1 || mmax < 1 || mmax > 64) return false;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)jacobi_eigh_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    hipLaunchKernelGGL(jacobi_eigh_kernel<true>, dim3(nb), dim3(TFJ_THREADS), lds_bytes(mmax, true, V0 != nullptr), st, mmax, W, vals,
                       (double *)nullptr, V0, Vkeep, 40, info, d_sizes, stride, vstride);
    *err = hipGetLastError();
    return *err == hipSuccess;
}

}  // namespace tfjac
