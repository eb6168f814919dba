// tf_refine.hip.h -- eigenvector refinement for the small matrices of the SCF cycle (n <= 64) in ONE launch, everything in LDS.
// Same algorithm as tfscf::ref_refine (tf_scf.hip.h: Newton-Schulz orthonormalisation + first-order rotation, after Ogita &
// Aishima 2018), but the five matrix products of a step run inside one workgroup on LDS-resident matrices -- one 16 x 16 tile per
// wave on the FP64 matrix core (V_MFMA_F64_16X16X4_F64) -- and the convergence
// test stays on the device: a whole solve is one kernel and one 4-byte status read, against ~60 rotation rounds with a barrier each
// for a warm-started Jacobi sweep.  Role in the reference: np.linalg.eigh in diagonalise_Fock_matrix (scf:244), of which the SCF
// iteration only needs the projector on the n_occ lowest eigenvectors (scf:183-211).
#pragma once
#include <hip/hip_runtime.h>

namespace tfref {

// pairs inside the occupied or inside the virtual space closer than this are one cluster: not rotated (the projector does not care)
#ifndef TF_REF_CLUSTER
#define TF_REF_CLUSTER 1e-5
#endif
#ifndef TF_REF_INTRA
#define TF_REF_INTRA 0.05             // largest rotation applied inside the occupied or the virtual space
#endif
#ifndef TF_REF_INNER
#define TF_REF_INNER 0                // fixed-point sweeps of the occupied-virtual equations per step (0: the diagonal first-order formula
                                      // alone).  2-4 sweeps halve the steps of a solve (3.8 -> 1.9 on N2/cc-pVTZ) but not its time: a solve is
                                      // ~150 us of launch, load, projection and status read-back around ~10 us steps
#endif
#define TFR_THREADS 1024
#define TFR_NMAX 64

typedef double tfr_v4d __attribute__((ext_vector_type(4)));

// matrices are padded to NP = a multiple of 16 (the MFMA tile) and stored with row stride NP + 2: even (aligned pairs) and, for
// NP <= 64, never a multiple of 32 -- the operand reads below (16 rows x 4 columns per instruction) are then bank-conflict free
__host__ __device__ inline int tfr_np(int n) { return (n + 15) & ~15; }
__host__ __device__ inline int tfr_stride(int np) { return np + 2; }
inline size_t lds_bytes(int n)
{
    const int np = tfr_np(n);
    return ((size_t)4 * np * tfr_stride(np) + 2 * TFR_NMAX + 2 * TFR_THREADS) * sizeof(double);
}

// One 16 x 16 output tile per wave on the FP64 matrix core: V_MFMA_F64_16X16X4_F64 takes A[i = lane % 16][k = lane / 16] and
// B[k = lane / 16][j = lane % 16] and leaves D[i = 4 v + lane / 16][j = lane % 16] in accumulator v (checked on gfx950).
// C = beta D + alpha A B^T (NT) or alpha A B (NN); tile (I, J) of this wave, or nothing if the wave has none.
template <bool NT, bool TA = false>
__device__ __forceinline__ void mm_tile(double *C, const double *A, const double *B, double alpha, double beta, const double *D, int np, int ns,
                                        int I, int J, int lane, bool on)
{
    if (!on) return;
    const int r = lane & 15, q = lane >> 4;
    const double *ap = TA ? A + q * ns + 16 * I + r : A + (16 * I + r) * ns + q;         // TA: C = alpha A^T B
    const double *bp = NT ? B + (16 * J + r) * ns + q : B + q * ns + 16 * J + r;
    const int astep = TA ? 4 * ns : 4, bstep = NT ? 4 : 4 * ns;
    tfr_v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < np / 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[kk * astep], bp[kk * bstep], acc, 0, 0, 0);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int idx = (16 * I + 4 * v + q) * ns + 16 * J + r;
        C[idx] = (beta != 0.0 ? beta * D[idx] : 0.0) + alpha * acc[v];
    }
}

// A: symmetric n x n (global, row-major); X: rows = orthonormal approximate eigenvectors (global, in/out).
// status[0] = 1: converged (X, lam, wocc written; wocc[i] = 1 for the n_occ lowest), 0: not (X untouched); status[1] = steps taken.
// Fused form (Fao != nullptr): A = sym(Xo Fao Xo) with the symmetric orthogonaliser Xo = S^-1/2 is formed in LDS first, and after
// convergence the density P = occ * sym(C_occ C_occ^T), C_occ = Xo x_occ (scf:183-250), is written to Pout -- a whole
// "diagonalise the Fock matrix and rebuild the density" step of the SCF cycle in one launch.
__global__ __launch_bounds__(TFR_THREADS) void refine_lds_kernel(int n, int n_occ, const double *__restrict__ A, double *__restrict__ X,
                                                                 double *__restrict__ lam_out, double *__restrict__ wocc_out, int max_steps,
                                                                 int *__restrict__ status, const double *__restrict__ Fao,
                                                                 const double *__restrict__ Xo, double *__restrict__ Pout, double occ,
                                                                 const int *__restrict__ sizes = nullptr, const int *__restrict__ noccs = nullptr,
                                                                 long long stride = 0, int vstride = 0)
{
    // a batch (the symmetry blocks of a larger matrix, tf_scf.hip.h: ref_refine_blocks): workgroup b refines the sizes[b] vectors of
    // block b (compact, leading dimension sizes[b]) against its block of A, with noccs[b] of them occupied; status[2 b], [2 b + 1]
    if (sizes) {
        const long long off = (long long)blockIdx.x * stride;
        n = sizes[blockIdx.x]; n_occ = noccs[blockIdx.x];
        A += off; X += off; lam_out += (size_t)blockIdx.x * vstride; wocc_out += (size_t)blockIdx.x * vstride; status += 2 * blockIdx.x;
        if (n < 1) { if (threadIdx.x == 0) { status[0] = 1; status[1] = 0; } return; }
    }
    extern __shared__ double sm[];
    const int np = tfr_np(n), ns = tfr_stride(np);
    double *sX = sm, *sA = sX + np * ns, *sT1 = sA + np * ns, *sT2 = sT1 + np * ns;
    double *sLam = sT2 + np * ns, *sOcc = sLam + TFR_NMAX, *sRed = sOcc + TFR_NMAX;       // sRed: 2 * TFR_THREADS
    __shared__ int sFlag;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = np / 16;
    const bool on = w < nt * nt;
    const int I = on ? w / nt : 0, J = on ? w - I * nt : 0;
    // load; the padding dimensions are decoupled vectors far up in the virtual space
    for (int e = tid; e < np * np; e += TFR_THREADS) {
        const int i = e / np, j = e - i * np;
        const bool in = i < n && j < n;
        sX[i * ns + j] = in ? X[(size_t)i * n + j] : ((i == j) ? 1.0 : 0.0);
        if (Fao) {
            sT1[i * ns + j] = in ? Fao[(size_t)i * n + j] : 0.0;
            sT2[i * ns + j] = in ? Xo[(size_t)i * n + j] : 0.0;
        } else
            sA[i * ns + j] = in ? 0.5 * (A[(size_t)i * n + j] + A[(size_t)j * n + i]) : ((i == j) ? 1.0e6 : 0.0);
    }
    __syncthreads();
    if (Fao) {                                                 // A = sym(Xo^T F Xo); the padding stays decoupled
        mm_tile<false>(sA, sT1, sT2, 1.0, 0.0, sX, np, ns, I, J, lane, on);          // F Xo
        __syncthreads();
        mm_tile<false, true>(sT1, sT2, sA, 1.0, 0.0, sX, np, ns, I, J, lane, on);    // Xo^T (F Xo)
        __syncthreads();
        for (int e = tid; e < np * np; e += TFR_THREADS) {
            const int i = e / np, j = e - i * np;
            sA[i * ns + j] = (i < n && j < n) ? 0.5 * (sT1[i * ns + j] + sT1[j * ns + i]) : ((i == j) ? 1.0e6 : 0.0);
        }
        __syncthreads();
    }
    int steps = 0;
    bool ok = false;
    for (int step = 0; step < max_steps; ++step) {
        // X <- (3/2 I - 1/2 X X^T) X   (not before the first step: the stored vectors are orthonormal, every solve closes with this)
        if (step > 0) {
            mm_tile<true>(sT1, sX, sX, 1.0, 0.0, sX, np, ns, I, J, lane, on);
            __syncthreads();
            mm_tile<false>(sT2, sT1, sX, -0.5, 1.5, sX, np, ns, I, J, lane, on);
            __syncthreads();
            { double *t = sX; sX = sT2; sT2 = t; }
        }
        if (ok) break;                                         // that was the closing orthonormalisation
        // S = X A X^T
        mm_tile<false>(sT1, sX, sA, 1.0, 0.0, sX, np, ns, I, J, lane, on);
        __syncthreads();
        mm_tile<true>(sT2, sT1, sX, 1.0, 0.0, sX, np, ns, I, J, lane, on);
        __syncthreads();
        if (tid < np) sLam[tid] = sT2[tid * ns + tid];
        __syncthreads();
        {                                                      // the n_occ lowest are occupied (ties by index, like a stable sort):
            const int i = tid >> 4, t16 = tid & 15;            // 16 lanes share the rank count of vector i
            int rank = 0;
            if (i < np) {
                const double li = sLam[i];
                for (int j = t16; j < np; j += 16) { const double lj = sLam[j]; rank += (lj < li || (lj == li && j < i)) ? 1 : 0; }
            }
            rank += __shfl_xor(rank, 8, 16); rank += __shfl_xor(rank, 4, 16); rank += __shfl_xor(rank, 2, 16); rank += __shfl_xor(rank, 1, 16);
            if (i < np && t16 == 0) sOcc[i] = rank < n_occ ? 1.0 : 0.0;
        }
        __syncthreads();
        // E (antisymmetric) into T1; largest rotation overall and between occupied and virtual vectors
        double emax = 0.0, eov = 0.0;
        for (int e = tid; e < np * np; e += TFR_THREADS) {
            const int i = e / np, j = e - i * np;
            double v = 0.0;
            if (i != j) {
                const double sij = 0.5 * (sT2[i * ns + j] + sT2[j * ns + i]);
                const double dl = sLam[j] - sLam[i];
                const bool ov = sOcc[i] != sOcc[j];
                if (ov || (fabs(sij) <= TF_REF_INTRA * fabs(dl) && fabs(dl) > TF_REF_CLUSTER)) v = sij / dl;
                emax = fmax(emax, fabs(v));
                if (ov) eov = fmax(eov, fabs(v));
            }
            sT1[i * ns + j] = v;
        }
#if TF_REF_INNER > 0
        // The formula above divides s_ia by (lambda_a - lambda_i): exact to first order only while the occupied-occupied and the
        // virtual-virtual blocks of S are diagonal.  They are not after a few Fock builds (the clusters inside a space are left
        // alone on purpose), and the iteration then converges linearly, a digit per step.  The occupied-virtual rotations that
        // annihilate s_ia to first order solve   sum_b e_ib s_ba - sum_j s_ij e_ja = s_ia   (b virtual, j occupied): a few Jacobi
        // sweeps with the diagonal as the preconditioner cost O(n_occ n_virt n) each -- nothing beside the five n^3 products of a
        // step.  16 lanes per entry (i, a) share the sum over m; new values go through a compact [n_occ][n_virt] buffer.
        {
            double *eNew = sRed;                                               // [n_occ x n_virt] <= 1024 doubles
            int *oList = reinterpret_cast<int *>(sRed + TFR_THREADS), *vList = oList + TFR_NMAX;
            __syncthreads();                                                   // E (first order) complete in T1
            if (tid < 64) {                                                    // the occupied / virtual vectors, in index order (np <= 64: wave 0)
                const bool in = tid < np, occ = in && sOcc[tid] == 1.0;
                const unsigned long long mo = __ballot(occ), below = (tid == 0) ? 0ull : (~0ull >> (64 - tid));
                const int ro = __popcll(mo & below);
                if (in) { if (occ) oList[ro] = tid; else vList[tid - ro] = tid; }
            }
            for (int e = tid; e < np * np; e += TFR_THREADS) {                 // S <- sym(S) in place (pairs i < j)
                const int i = e / np, j = e - i * np;
                if (i < j) { const double m2 = 0.5 * (sT2[i * ns + j] + sT2[j * ns + i]); sT2[i * ns + j] = m2; sT2[j * ns + i] = m2; }
            }
            __syncthreads();
            const int no = n_occ, nv = np - n_occ, npair = no * nv;
            const int grp = tid >> 4, t16 = tid & 15;
            bool mocc[TFR_NMAX / 16];
#pragma unroll
            for (int u = 0; u < TFR_NMAX / 16; ++u) mocc[u] = (t16 + 16 * u < np) && sOcc[t16 + 16 * u] == 1.0;
            for (int sweep = 0; sweep < TF_REF_INNER; ++sweep) {
                for (int q = grp; q < npair; q += TFR_THREADS / 16) {
                    const int qo = q / nv, i = oList[qo], a2 = vList[q - qo * nv];
                    double r = 0.0;
#pragma unroll
                    for (int u = 0; u < TFR_NMAX / 16; ++u) {
                        const int m = t16 + 16 * u;
                        if (m < np && m != i && m != a2)
                            r += mocc[u] ? sT2[i * ns + m] * sT1[m * ns + a2] : -sT1[i * ns + m] * sT2[m * ns + a2];
                    }
                    r += __shfl_xor(r, 8, 16); r += __shfl_xor(r, 4, 16); r += __shfl_xor(r, 2, 16); r += __shfl_xor(r, 1, 16);
                    if (t16 == 0) eNew[q] = (sT2[i * ns + a2] + r) / (sLam[a2] - sLam[i]);
                }
                __syncthreads();
                for (int q = tid; q < npair; q += TFR_THREADS) {
                    const int qo = q / nv, i = oList[qo], a2 = vList[q - qo * nv];
                    const double v = eNew[q];
                    sT1[i * ns + a2] = v; sT1[a2 * ns + i] = -v;
                }
                __syncthreads();
            }
            emax = 0.0; eov = 0.0;
            for (int e = tid; e < np * np; e += TFR_THREADS) {
                const int i = e / np, j = e - i * np;
                const double v = fabs(sT1[i * ns + j]);
                emax = fmax(emax, v);
                if (sOcc[i] != sOcc[j]) eov = fmax(eov, v);
            }
            __syncthreads();                                                   // (sRed is reused by the reduction below)
        }
#endif
        // block maxima: inside a wave by shuffles, across the 16 waves through LDS (two barriers instead of a ten-level tree)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { emax = fmax(emax, __shfl_xor(emax, d, 64)); eov = fmax(eov, __shfl_xor(eov, d, 64)); }
        if (lane == 0) { sRed[w] = emax; sRed[TFR_THREADS + w] = eov; }
        __syncthreads();
        emax = sRed[0]; eov = sRed[TFR_THREADS];
#pragma unroll
        for (int u = 1; u < TFR_THREADS / 64; ++u) { emax = fmax(emax, sRed[u]); eov = fmax(eov, sRed[TFR_THREADS + u]); }
        __syncthreads();
        if (!(emax <= 0.3)) break;                             // NaN, or not in the quadratic regime: the caller diagonalises
        // X <- X + E^T X = X - E X
        mm_tile<false>(sT2, sT1, sX, -1.0, 1.0, sX, np, ns, I, J, lane, on);
        __syncthreads();
        { double *t = sX; sX = sT2; sT2 = t; }
        ++steps;
        ok = eov < 1e-9 && emax < 0.1;       // rotations inside the occupied or the virtual space leave the projector alone
    }
    if (tid == 0) sFlag = ok ? 1 : 0;
    __syncthreads();
    if (sFlag) {
        for (int e = tid; e < n * n; e += TFR_THREADS) { const int i = e / n, j = e - i * n; X[e] = sX[i * ns + j]; }
        if (tid < n) { lam_out[tid] = sLam[tid]; wocc_out[tid] = sOcc[tid]; }
        if (Pout) {                                            // (uniform branch: sFlag and Pout are the same for every thread)
            // rows of the occupied eigenvectors (others zeroed) -> AO basis: T = Xocc Xo^T; P = occ * sym(T^T T)
            for (int e = tid; e < np * np; e += TFR_THREADS) {
                const int i = e / np, j = e - i * np;
                sT1[i * ns + j] = (i < n && j < n) ? sOcc[i] * sX[i * ns + j] : 0.0;
                sA[i * ns + j] = (i < n && j < n) ? Xo[(size_t)i * n + j] : 0.0;
            }
            __syncthreads();
            mm_tile<true>(sT2, sT1, sA, 1.0, 0.0, sX, np, ns, I, J, lane, on);
            __syncthreads();
            mm_tile<false, true>(sT1, sT2, sT2, occ, 0.0, sX, np, ns, I, J, lane, on);
            __syncthreads();
            for (int e = tid; e < n * n; e += TFR_THREADS) { const int i = e / n, j = e - i * n; Pout[e] = 0.5 * (sT1[i * ns + j] + sT1[j * ns + i]); }
        }
    }
    if (tid == 0) { status[0] = sFlag; status[1] = steps; }
}

// n <= 64: the Fock matrix and the DIIS error of an SCF iteration in ONE launch (everything in LDS, products on the FP64 matrix core):
//     F = sym(H + J - hfx/2 K [+ V_XC])                                   (scf:497-531)
//     e = X^T (F P S - S P F) X,  with G = F P S and S P F = G^T            (scf:906-920; F, P, S symmetric)
// F goes to Fout and to the DIIS history slot Fhist, e to Eout.  Replaces nine launches (two element-wise kernels, six rocBLAS GEMMs, a
// copy) whose host side, not their arithmetic, is what an iteration at this size consists of.
__global__ __launch_bounds__(TFR_THREADS) void fock_diis_lds_kernel(int n, const double *__restrict__ H, const double *__restrict__ J,
                                                                    const double *__restrict__ K, double hfx, const double *__restrict__ Vxc,
                                                                    const double *__restrict__ P, const double *__restrict__ S,
                                                                    const double *__restrict__ X, double *__restrict__ Fout,
                                                                    double *__restrict__ Fhist, double *__restrict__ Eout)
{
    extern __shared__ double sm[];
    const int np = tfr_np(n), ns = tfr_stride(np);
    double *sA = sm, *sB = sA + np * ns, *sT1 = sB + np * ns, *sT2 = sT1 + np * ns;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = np / 16;
    const bool on = w < nt * nt;
    const int I = on ? w / nt : 0, Jt = on ? w - I * nt : 0;
    auto fock = [&](int i, int j) {
        const size_t e = (size_t)i * n + j;
        double f = H[e] + J[e] - (1.0 / 2.0) * K[e] * hfx;
        if (Vxc) f += Vxc[e];
        return f;
    };
    for (int e = tid; e < np * np; e += TFR_THREADS) {
        const int i = e / np, j = e - i * np;
        const bool in = i < n && j < n;
        const double f = in ? 0.5 * (fock(i, j) + fock(j, i)) : 0.0;
        sA[i * ns + j] = f;
        sB[i * ns + j] = in ? P[(size_t)i * n + j] : 0.0;
        if (in) { Fout[(size_t)i * n + j] = f; Fhist[(size_t)i * n + j] = f; }
    }
    __syncthreads();
    mm_tile<false>(sT1, sA, sB, 1.0, 0.0, sA, np, ns, I, Jt, lane, on);              // F P
    __syncthreads();
    for (int e = tid; e < np * np; e += TFR_THREADS) { const int i = e / np, j = e - i * np; sB[i * ns + j] = (i < n && j < n) ? S[(size_t)i * n + j] : 0.0; }
    __syncthreads();
    mm_tile<false>(sT2, sT1, sB, 1.0, 0.0, sA, np, ns, I, Jt, lane, on);             // G = F P S
    __syncthreads();
    for (int e = tid; e < np * np; e += TFR_THREADS) {
        const int i = e / np, j = e - i * np;
        sT1[i * ns + j] = sT2[i * ns + j] - sT2[j * ns + i];                          // F P S - S P F
        sB[i * ns + j] = (i < n && j < n) ? X[(size_t)i * n + j] : 0.0;
    }
    __syncthreads();
    mm_tile<false, true>(sT2, sB, sT1, 1.0, 0.0, sA, np, ns, I, Jt, lane, on);       // X^T (.)
    __syncthreads();
    mm_tile<false>(sT1, sT2, sB, 1.0, 0.0, sA, np, ns, I, Jt, lane, on);             // (.) X
    __syncthreads();
    for (int e = tid; e < n * n; e += TFR_THREADS) { const int i = e / n, j = e - i * n; Eout[e] = sT1[i * ns + j]; }
}

inline bool launch_fock_diis(int n, const double *H, const double *J, const double *K, double hfx, const double *Vxc, const double *P,
                             const double *S, const double *X, double *Fout, double *Fhist, double *Eout, hipStream_t st, hipError_t *err)
{
    static bool attr_set = false;
    if (n < 2 || n > TFR_NMAX) return false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)fock_diis_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    const int np = tfr_np(n);
    hipLaunchKernelGGL(fock_diis_lds_kernel, dim3(1), dim3(TFR_THREADS), (size_t)4 * np * tfr_stride(np) * sizeof(double), st, n, H, J, K, hfx, Vxc, P,
                       S, X, Fout, Fhist, Eout);
    *err = hipGetLastError();
    return *err == hipSuccess;
}

inline bool launch(int n, int n_occ, const double *A, double *X, double *lam, double *wocc, int *status, hipStream_t st, hipError_t *err,
                   const double *Fao = nullptr, const double *Xo = nullptr, double *Pout = nullptr, double occ = 2.0)
{
    static bool attr_set = false;
    if (n < 2 || n > TFR_NMAX) return false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)refine_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    hipLaunchKernelGGL(refine_lds_kernel, dim3(1), dim3(TFR_THREADS), lds_bytes(n), st, n, n_occ, A, X, lam, wocc, 12, status, Fao, Xo, Pout, occ);
    *err = hipGetLastError();
    return *err == hipSuccess;
}

// a batch of nb blocks of at most mmax <= 64 vectors each in one launch (non-fused form)
inline bool launch_batch(int nb, int mmax, const int *d_sizes, const int *d_noccs, const double *A, double *X, long long stride, double *lam,
                         double *wocc, int vstride, int *status, hipStream_t st, hipError_t *err)
{
    static bool attr_set = false;
    if (nb < 1 || mmax < 1 || mmax > TFR_NMAX) return false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)refine_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    hipLaunchKernelGGL(refine_lds_kernel, dim3(nb), dim3(TFR_THREADS), lds_bytes(mmax), st, mmax, 0, A, X, lam, wocc, 12, status,
                       (const double *)nullptr, (const double *)nullptr, (double *)nullptr, 2.0, d_sizes, d_noccs, stride, vstride);
    *err = hipGetLastError();
    return *err == hipSuccess;
}

}  // namespace tfref
