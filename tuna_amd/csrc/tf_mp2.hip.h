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

// ---- hand-written first quarter on the packed rows (round 3) ---------------------------------------------------------------------
// R[(mu nu)][sigma][p] = sum_lambda C3[lambda][p] Lsym[(mu nu)][lambda][sigma]   for the n3 <= 32 columns of C3 (the occupied orbitals of
// an (ia|jb) transformation: tuna_ci.py:204-255 contracts a full index first, 2 N^5; contracting the SHORT index first leaves 2 o N^4
// and a tensor 1/(N/o) the size).  Lsym = the stored part of the row as a symmetric matrix (unpack_own_rows_kernel above: pairs
// (lambda sigma) <= (mu nu), the pair equal to (mu nu) halved).  The kernel reads the packed segments themselves -- 7 GB at N = 400
// instead of 26 GB of expanded blocks -- on the FP64 matrix core (v_mfma_f64_16x16x4_f64), C3 as the B operand:
//   * one workgroup (4 waves) per stored row; the output AOs sigma are cut into blocks of 16 of one parity class x, a block belongs to
//     ONE wave, which accumulates both images of the stored triangle into the block's 16 x n3 tile:
//       "column" image  R[l][p] += sum_k V[k][l] C3[k][p]   (k of class x ^ c walks the segments that reach the block's columns l),
//       "row" image     R[k][p] += sum_l V[k][l] C3[l][p]   (the block's own 16 segments k, all their columns l; l == k skipped);
//     so every value is loaded twice (the second time from L2: a row is ~90 KB) and nothing is ever merged between waves: no LDS,
//     no atomics, one store per output element, bitwise reproducible;
//   * operands go from global memory straight into the MFMA lane layout (A[row = lane & 15][k = lane >> 4]): the column image reads
//     four segments x 128 contiguous bytes per instruction, the row image 16 segments x 64-byte runs (eight K steps per four loads);
//   * the tile is written twice, to R[mu][nu] and R[nu][mu] (ORIGINAL bra indices, internal sigma), so that the bra half is three
//     plain rocBLAS GEMMs on contiguous matrices (transform_q1 below).
typedef double tfm_v4d __attribute__((ext_vector_type(4)));

struct Q1Args {
    const double *eri;
    const long long *rowoff;
    const int *rowsec;
    const int2 *row_ij;            // original (i >= j) of every local row
    const double *C3p;             // [N internal][16 NT]: C3 in internal AO order, columns padded with zeros
    double *R;                     // [local row][N internal sigma][n3]
    int n3;
    int c3ld;                      // leading dimension of C3p (16 x column tiles of the whole n3)
    int dbg;                       // timing experiments (TF_Q1_DBG bits: 1 no mirrored store, 2 no store, 4 no column image, 8 no row image)
};

// BLDS: C3 is staged in LDS ([N][n3r], n3r = n3 rounded up to even) once per workgroup, which then walks `rpw` consecutive rows -- the
// B operands of every MFMA are then LDS reads; from global memory (L2) they were two thirds of the bytes through the texture
// addresser, which ran at ~75 % for the four waves of a CU.  Without (N n3r doubles do not fit): B operands from C3p in global memory.
// The rows of a workgroup (`rpw` consecutive ones) are prepared together -- per row a header, the segment table and, per block, the first
// segment that reaches it -- and the (row, block) items then go through a work queue (an LDS counter): a wave takes the next item when
// it is free, whatever row it belongs to (every fourth block of a row per wave left the slowest wave 20 % behind the mean; and no
// barrier separates the rows any more).  Which wave computes an item does not change its result: one owner, fixed order inside.
#define TFQ1_RPW 4
#ifndef TFQ1_THREADS
#define TFQ1_THREADS 512            // eight waves share one staging of C3 (two workgroups per CU by LDS: four waves per SIMD)
#endif
struct Q1Row { long long rowoff; int i, j, c, iI, lamj, pad; };   // (R row = the local row number: r_first + position in the workgroup)
// NX > 0 (with NT = 1, BLDS): the columns 16 .. 16 + NX - 1 of C3 (an occupied space of 17 - 20 orbitals: Ar2 has 18) do not get a second MFMA
// column tile, of which they would fill an eighth -- NX multiply-adds per loaded value on the vector ALU instead (the C3 element is a
// broadcast LDS read; the partial sums of a lane's share of the K index are added up over the four lane groups at the end of the block).
template <int NT, bool BLDS, int NX = 0>
__global__ __launch_bounds__(TFQ1_THREADS, 4) void mo_q1_kernel(Q1Args Q, BLayout L, int n3r, int rpw, long long n_rows, int nblk)
{
    extern __shared__ double sQ1[];
    const int N = L.N;
    // LDS: [rpw][N] segment tables | [rpw] row headers | [rpw][nblk] block starts | counter | C3 [N][n3r] (BLDS)
    int2 *sSegAll = reinterpret_cast<int2 *>(sQ1);                  // per row and internal AO k: offset of its segment in the row's unit; columns it holds
    Q1Row *sRow = reinterpret_cast<Q1Row *>(sSegAll + (size_t)TFQ1_RPW * N);
    int *sLo = reinterpret_cast<int *>(sRow + TFQ1_RPW);            // [rpw][nblk]: first member of the column walk of block b
    int *sCounter = sLo + TFQ1_RPW * nblk;
    double *sC3 = reinterpret_cast<double *>(sCounter + 2 + ((TFQ1_RPW * nblk) & 1));   // 8-byte aligned: sLo holds rpw * nblk ints, + 2 for the counter
    const int lane = threadIdx.x & 63;
    const int m = lane & 15, kk = lane >> 4;
    constexpr int NP = 16 * NT;
    if (BLDS) {
        for (int e = threadIdx.x; e < N * n3r; e += TFQ1_THREADS) { const int x = e / n3r, p = e - x * n3r; sC3[e] = Q.C3p[(size_t)x * Q.c3ld + p]; }
    }
    bool colok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) colok[t] = 16 * t + m < n3r;
    const long long r_first = (long long)blockIdx.x * rpw;
    const int nrow = (int)min((long long)rpw, n_rows - r_first);
    int bfirst[5];
    bfirst[0] = 0;
#pragma unroll
    for (int x = 0; x < 4; ++x) bfirst[x + 1] = bfirst[x] + (L.itab[BL_CSIZE + x] + 15) / 16;
    if (threadIdx.x < nrow) {
        const long long r = r_first + threadIdx.x;
        const int2 ij = Q.row_ij[r];
        const int wi = L.ao[ij.x], wj = L.ao[ij.y];
        sRow[threadIdx.x] = Q1Row{Q.rowoff[r], ij.x, ij.y, ao_cls(wi) ^ ao_cls(wj), ao_sigma(L, wi), ao_loc(wj), 0};
    }
    if (threadIdx.x == 0) *sCounter = 0;
    __syncthreads();
    // segment tables: AOs beyond i (original order) hold nothing, the segment of k == i ends at l == j
    for (int e = threadIdx.x; e < nrow * N; e += TFQ1_THREADS) {
        const int rr = e / N, kI = e - rr * N;
        const Q1Row R = sRow[rr];
        const int *rs = Q.rowsec + 6 * (size_t)(r_first + rr);
        const int a = L.clsI[kI];
        const KInfo ki = L.kinfo[(size_t)R.c * N + kI];
        const int pc = (ki.cnt + TF_SEG_PAD - 1) & ~(TF_SEG_PAD - 1);
        const bool have = kI - bl_cstart(L, a) < L.cntA[(size_t)a * N + R.iI];
        sSegAll[e] = make_int2(rs[5] * (rs[a] + ki.offA) + rs[4] * pc, have ? (kI == R.iI ? R.lamj + 1 : ki.cnt) : 0);
    }
    __syncthreads();
    // per block: the first member of class a = x ^ c whose segment is longer than s0 (cnt is non-decreasing, except that the segment of
    // k == i -- the last member when i is of that class -- is cut at l == j: it stays out of the search and is tested like every element)
    for (int e = threadIdx.x; e < nrow * nblk; e += TFQ1_THREADS) {
        const int rr = e / nblk, blk = e - rr * nblk;
        const Q1Row R = sRow[rr];
        const int x = blk >= bfirst[3] ? 3 : (blk >= bfirst[2] ? 2 : (blk >= bfirst[1] ? 1 : 0));
        const int s0 = 16 * (blk - (x == 3 ? bfirst[3] : (x == 2 ? bfirst[2] : (x == 1 ? bfirst[1] : 0))));
        const int a = x ^ R.c, a0 = bl_cstart(L, a);
        const int klim = L.cntA[(size_t)a * N + R.iI];
        const int2 *sg = sSegAll + (size_t)rr * N;
        int lo = 0, hi = (klim > 0 && a0 + klim - 1 == R.iI) ? klim - 1 : klim;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sg[a0 + mid].y > s0) hi = mid; else lo = mid + 1; }
        sLo[e] = lo;
    }
    __syncthreads();
    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(sCounter, 1);
        item = __shfl(item, 0, 64);
        if (item >= nrow * nblk) break;
        const int rr = __builtin_amdgcn_readfirstlane(item / nblk), blk = __builtin_amdgcn_readfirstlane(item - (item / nblk) * nblk);
        const Q1Row R = sRow[rr];
        const int c = R.c, iI = R.iI, lamj = R.lamj;
        const double *__restrict__ T = Q.eri + R.rowoff;
        const int2 *sSeg = sSegAll + (size_t)rr * N;
        const int x = blk >= bfirst[3] ? 3 : (blk >= bfirst[2] ? 2 : (blk >= bfirst[1] ? 1 : 0));
        const int s0 = 16 * (blk - (x == 3 ? bfirst[3] : (x == 2 ? bfirst[2] : (x == 1 ? bfirst[1] : 0))));
        const int nx = L.itab[BL_CSIZE + x], x0 = bl_cstart(L, x);
        const int a = x ^ c, a0 = bl_cstart(L, a), na = L.itab[BL_CSIZE + a];
        const int lo_col = sLo[item];
        tfm_v4d acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = tfm_v4d{0.0, 0.0, 0.0, 0.0};
        double accx[NX > 0 ? NX : 1];
#pragma unroll
        for (int q = 0; q < (NX > 0 ? NX : 1); ++q) accx[q] = 0.0;
        // Both images run as software pipelines: the loads of the next pass are issued before the MFMAs of the current one, and the first
        // pass of the row image before the column image starts (a wave has one or two partners on its SIMD, not enough to hide an HBM miss
        // behind 16 MFMAs by occupancy alone).
        // ---- row image: the block's own segments k = s0 .. s0 + 15 of class x, all their columns l of class x ^ c, l != k
        const int klimR = L.cntA[(size_t)x * N + iI];
        const bool have_row = s0 < klimR && !(Q.dbg & 8);
        const int klR = s0 + m;
        const bool vkR = have_row && klR < klimR;
        const int kIR = x0 + (vkR ? klR : 0);
        const int2 sgR = sSeg[kIR];
        const int pcR = vkR ? ((sgR.y + TF_SEG_PAD - 1) & ~(TF_SEG_PAD - 1)) : 0;   // (the slots between cnt and the pad hold zeros; so do those beyond l == j of k == i)
        const double *__restrict__ segR = T + sgR.x;
        int cmax = 0;                                                                // the longest segment of the block: its last one, or -- the
        if (have_row) {                                                              // segment of k == i being cut at l == j -- the one before
            const int last = min(s0 + 15, klimR - 1);
            cmax = sSeg[x0 + last].y;
            if (last > s0) cmax = max(cmax, sSeg[x0 + last - 1].y);
        }
        const bool diag_cls = c == 0;                                                // l and k are of one class: l == k <=> lam == kl
        auto row_load = [&](int l0, double (&v)[8]) {
            const int lb = l0 + 8 * kk;                                              // this lane's eight columns
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = lb + 2 * u < pcR;
                const double2 t2 = in ? *reinterpret_cast<const double2 *>(segR + lb + 2 * u) : make_double2(0.0, 0.0);
                v[2 * u] = t2.x; v[2 * u + 1] = t2.y;
            }
        };
        auto row_mma = [&](int l0, double (&v)[8]) {
            const int lb = l0 + 8 * kk;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int lam = lb + u;
                if (lam >= sgR.y || (diag_cls && lam == klR)) v[u] = 0.0;
                if (kIR == iI && lam == lamj) v[u] *= 0.5;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int lam = lb + u;
                const bool vl = lam < na;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const double b = BLDS ? ((vl && colok[t]) ? sC3[(a0 + (vl ? lam : 0)) * n3r + 16 * t + m] : 0.0)
                                          : (vl ? Q.C3p[(size_t)(a0 + (vl ? lam : 0)) * Q.c3ld + 16 * t + m] : 0.0);
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], b, acc[t], 0, 0, 0);
                }
                if (NX > 0 && vl) {
#pragma unroll
                    for (int q = 0; q < NX; ++q) accx[q] += v[u] * sC3[(a0 + lam) * n3r + 16 + q];
                }
            }
        };
        double rA[8], rB[8];
        if (cmax > 0) row_load(0, rA);
        // ---- column image: the segments k of class a = x ^ c that reach the columns s0 .. s0 + 15 of class x; a lane takes eight
        //      consecutive k per pass (eight loads in flight, then eight K steps)
        if (!(Q.dbg & 4)) {
            const int klim = L.cntA[(size_t)a * N + iI];
            const int lo = lo_col;
            const int lam = s0 + m;
            auto col_load = [&](int kl0, double (&v)[8], int (&kIs)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kl = kl0 + 8 * kk + u;
                    const bool vk = kl < klim;
                    const int kI = a0 + (vk ? kl : 0);
                    const int2 sg = sSeg[kI];
                    const bool ok = vk && lam < sg.y;
                    v[u] = ok ? __builtin_nontemporal_load(T + sg.x + lam) : 0.0;
                    kIs[u] = vk ? kI : -1;
                }
            };
            auto col_mma = [&](double (&v)[8], const int (&kIs)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (kIs[u] == iI && lam == lamj) v[u] *= 0.5;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const double b = BLDS ? ((kIs[u] >= 0 && colok[t]) ? sC3[kIs[u] * n3r + 16 * t + m] : 0.0)
                                              : (kIs[u] >= 0 ? Q.C3p[(size_t)kIs[u] * Q.c3ld + 16 * t + m] : 0.0);
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], b, acc[t], 0, 0, 0);
                    }
                    if (NX > 0 && kIs[u] >= 0) {
#pragma unroll
                        for (int q = 0; q < NX; ++q) accx[q] += v[u] * sC3[kIs[u] * n3r + 16 + q];
                    }
                }
            };
            int kl0 = lo & ~7;
            if (kl0 < klim) {
                double vA[8], vB[8];
                int kA[8], kB[8];
                col_load(kl0, vA, kA);
                for (;;) {
                    const int k1 = kl0 + 32;
                    if (k1 < klim) col_load(k1, vB, kB);
                    col_mma(vA, kA);
                    if (k1 >= klim) break;
                    const int k2 = k1 + 32;
                    if (k2 < klim) col_load(k2, vA, kA);
                    col_mma(vB, kB);
                    if (k2 >= klim) break;
                    kl0 = k2;
                }
            }
        }
        if (cmax > 0) {
            int l0 = 0;
            for (;;) {
                const int l1 = l0 + 32;
                if (l1 < cmax) row_load(l1, rB);
                row_mma(l0, rA);
                if (l1 >= cmax) break;
                const int l2 = l1 + 32;
                if (l2 < cmax) row_load(l2, rA);
                row_mma(l1, rB);
                if (l2 >= cmax) break;
                l0 = l2;
            }
        }
        if (NX > 0) {
            // the extra columns: lane (m, kk) holds the part of row sigma = x0 + s0 + m that its share of the K index gave; the four lane groups
            // are added in fixed order (xor 16, xor 32) and the group kk == 0 writes
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                double t = accx[q];
                t += __shfl_xor(t, 16, 64);
                t += __shfl_xor(t, 32, 64);
                const int row = s0 + m;
                if (kk == 0 && row < nx) {
                    const size_t sig = (size_t)(x0 + row);
                    if (!(Q.dbg & 2)) Q.R[((size_t)(r_first + rr) * N + sig) * Q.n3 + 16 + q] = t;
                }
            }
        }
        // ---- the tile: rows sigma = x0 + s0 + 4 v + (lane >> 4), columns p = 16 t + (lane & 15); both bra orders
#pragma unroll
        for (int vv = 0; vv < 4; ++vv) {
            const int row = s0 + 4 * vv + kk;
            if (row >= nx) continue;
            const size_t sig = (size_t)(x0 + row);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int p = 16 * t + m;
                if (p >= Q.n3) continue;
                const double val = acc[t][vv];
                if (!(Q.dbg & 2)) Q.R[((size_t)(r_first + rr) * N + sig) * Q.n3 + p] = val;
            }
        }
    }
}

// Cp[x][0 .. NP) = C[origI[x]][0 .. n) padded with zeros: coefficient rows in internal AO order, MFMA column tiles
__global__ void permute_rows_padded_kernel(const double *__restrict__ C, const int *__restrict__ origI, int N, int n, int NP, double *__restrict__ Cp)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * NP) return;
    const int x = e / NP, p = e - x * NP;
    Cp[e] = p < n ? C[(size_t)origI[x] * n + p] : 0.0;
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
                     const double *dC4, int n4, double *d_out, double *gemm_seconds, std::string &msg, const TView *tiles = nullptr)
{
    int rc = TF_OK;
    const long long n34 = (long long)n3 * n4;
    const long long row_len = (long long)N * ld;
    double *dR = nullptr, *dQ = nullptr, *dQfull = nullptr, *dW = nullptr, *dM = nullptr, *dC3i = nullptr, *dC4i = nullptr;
    const bool packed = d_rowoff != nullptr || tiles != nullptr;      // rows that hold the pairs (kl) <= (ij) only
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
                if (tiles)
                    hipLaunchKernelGGL(unpack_own_rows_blocked_tiles_kernel, dim3((unsigned)((PR.NP[c] + 255) / 256), (unsigned)nb), dim3(256), 0, 0, d_eri,
                                       *tiles, BL, d_row_ij, PR.d_class_rows + q0, c, RB, dM);
                else
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

// T[p][nu][y] = sum_mu C1[mu][p] R[row(mu, nu)][y]  (p < n1 <= 32, y < Y = N n3): the bra index mu -> occupied orbital.  R holds one block
// of Y doubles per STORED row (mu >= nu, this rank's); the kernel follows the row map -- row(mu, nu) = row(nu, mu), absent rows (another
// rank's) contribute nothing -- so the first quarter writes every block once (4.6 GB at N = 400 instead of 9.2 for both bra orders).
// rocBLAS took a 128 x 128 macro tile for the 18-wide output and ran compute-bound on the padding (4.6 ms); here the orbitals are the M
// dimension of v_mfma_f64_16x16x4_f64 (one or two tiles), 64 consecutive y per wave the N dimension (four tiles: 512 contiguous bytes of
// every block), C1 (original AO order) and the row numbers of the workgroup's nu in LDS, the next K step's loads in flight.
// Grid: (ceil(Y / (64 waves)), N); a workgroup works on ONE nu.
#define TFB1_THREADS 384          // six waves share one staging of C1: two workgroups per CU by LDS = three waves per SIMD (157 VGPRs)
template <int MT>
__global__ __launch_bounds__(TFB1_THREADS, 3) void mo_bra1_kernel(const double *__restrict__ R, const double *__restrict__ C1, const int *__restrict__ rowmap,
                                                                 BLayout L, int n1, int Y, double *__restrict__ T)
{
    extern __shared__ double sB1[];
    const int N = L.N;
    const int n1r = (n1 + 1) & ~1;
    double *sC1 = sB1;                                              // [N][n1r], rows = ORIGINAL AO index mu
    int *sRowOf = reinterpret_cast<int *>(sB1 + (size_t)N * n1r);    // [N]: local row of (mu, nu), -1 = not on this rank
    const int nu = blockIdx.y;
    for (int e = threadIdx.x; e < N * n1r; e += TFB1_THREADS) { const int mu = e / n1r, p = e - mu * n1r; sC1[e] = p < n1 ? C1[(size_t)mu * n1 + p] : 0.0; }
    {
        const int snu = ao_sigma(L, L.ao[nu]);
        for (int mu = threadIdx.x; mu < N; mu += TFB1_THREADS) {
            const int smu = ao_sigma(L, L.ao[mu]);
            const int hi = max(smu, snu), lo = min(smu, snu);
            sRowOf[mu] = rowmap[(size_t)hi * (hi + 1) / 2 + lo];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int m = lane & 15, kk = lane >> 4;
    const int y0 = (blockIdx.x * (TFB1_THREADS / 64) + w) * 64;
    if (y0 >= Y) return;
    bool rowok[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) rowok[t] = 16 * t + m < n1r;
    tfm_v4d acc[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = tfm_v4d{0.0, 0.0, 0.0, 0.0};
    bool yok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) yok[j] = y0 + 16 * j + m < Y;
    const double *__restrict__ Ry = R + y0 + m;
    auto load = [&](int mu0, double (&b)[4]) {
        const int mu = mu0 + kk;
        const int r = mu < N ? sRowOf[mu] : -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (r >= 0 && yok[j]) ? __builtin_nontemporal_load(Ry + (size_t)r * Y + 16 * j) : 0.0;
    };
    auto mma = [&](int mu0, const double (&b)[4]) {
        const int mu = mu0 + kk;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const double a = (mu < N && rowok[t]) ? sC1[mu * n1r + 16 * t + m] : 0.0;       // A[row p = lane & 15][k = mu]
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[j], acc[t][j], 0, 0, 0);
        }
    };
    double bA[4], bB[4];
    load(0, bA);
    for (int mu0 = 0; mu0 < N; mu0 += 8) {                            // two buffers: the loads of the next K step are in flight
        if (mu0 + 4 < N) load(mu0 + 4, bB);
        mma(mu0, bA);
        if (mu0 + 4 >= N) break;
        if (mu0 + 8 < N) load(mu0 + 8, bA);
        mma(mu0 + 4, bB);
    }
    // D[row p = 4 v + (lane >> 4)][col y = lane & 15]
    const size_t X = (size_t)N * Y;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int p = 16 * t + 4 * v + kk;
            if (p >= n1) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (yok[j]) T[(size_t)p * X + (size_t)nu * Y + y0 + 16 * j + m] = acc[t][j][v];
        }
}

// The packed-layout transformation with the short index first (n3 <= 32): first quarter by mo_q1_kernel, then three rocBLAS GEMMs
//     T [p1][nu][sigma][p3]  = sum_mu C1[mu][p1] R[mu][nu][sigma][p3]                 (one GEMM, K = N)
//     T2[p1][p2][sigma][p3]  = sum_nu C2[nu][p2] T[p1][nu][sigma][p3]                 (batched over p1)
//     out[p1][p2][p3][p4]    = sum_sigma T2[p1][p2][sigma][p3] C4[sigma][p4]          (batched over (p1, p2))
// on a caller-owned pool of q1_pool_doubles() doubles (kept by the context: no allocation per call).  Same contract as transform():
// d_out is the transform of L (the stored part, its diagonal halved); rows the rank does not own are absent from the row map and
// contribute nothing.  Condition (checked by the caller): n1 <= 32 and n3 <= 32.
inline size_t q1_pool_doubles(int N, long long n_rows, int n1, int n2, int n3, int n4)
{
    const size_t NP = (size_t)16 * ((n3 + 15) / 16);
    const size_t Y = (size_t)N * n3;
    // R [rows][Y] (later T2 [n1][n2][Y] in the same place) | T [n1][N][Y] | C3 padded | C4 in internal order
    return std::max((size_t)std::max<long long>(1, n_rows) * Y, (size_t)n1 * n2 * Y) + (size_t)n1 * N * Y + (size_t)N * NP + (size_t)N * n4 + 64;
}

inline int transform_q1(rocblas_handle blas, const double *d_eri, const long long *d_rowoff, const int *d_rowsec, const BLayout &BL,
                        const int2 *d_row_ij, const int *d_rowmap, long long n_rows, int N, const double *dC1, int n1, const double *dC2, int n2,
                        const double *dC3, int n3, const double *dC4, int n4, double *d_out, double *pool, double *seconds, std::string &msg)
{
    int rc = TF_OK;
    const double one = 1.0, zero = 0.0;
    const int NT = (n3 + 15) / 16, NP = 16 * NT;
    const size_t X = (size_t)N * N * n3, Y = (size_t)N * n3;
    const size_t r_doubles = std::max((size_t)std::max<long long>(1, n_rows) * Y, (size_t)n1 * n2 * Y);
    double *dR = pool, *dT = dR + r_doubles, *dC3p = dT + (size_t)n1 * X, *dC4i = dC3p + (size_t)N * NP, *dT2 = dR;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (X > 0x7fffffffULL || (size_t)n1 * n2 > 0x7fffffffULL) { msg = "AO->MO transformation: dimension overflow"; return TF_EINVAL; }
    TFM_HIP(hipEventCreate(&e0));
    TFM_HIP(hipEventCreate(&e1));
    TFM_HIP(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(permute_rows_padded_kernel, dim3((unsigned)((N * NP + 255) / 256)), dim3(256), 0, 0, dC3, BL.origI, N, n3, NP, dC3p);
    hipLaunchKernelGGL(permute_rows_kernel, dim3((unsigned)((N * n4 + 255) / 256)), dim3(256), 0, 0, dC4, BL.origI, N, n4, dC4i);
    if (n_rows > 0) {
        const char *dbg = getenv("TF_Q1_DBG");
        Q1Args Q{d_eri, d_rowoff, d_rowsec, d_row_ij, dC3p, dR, n3, NP, dbg ? atoi(dbg) : 0};
        const int n3r = (n3 + 1) & ~1;
        int nblk = 0;
        {
            int csz[4];
            TFM_HIP(hipMemcpy(csz, BL.itab + BL_CSIZE, sizeof(csz), hipMemcpyDeviceToHost));
            for (int x = 0; x < 4; ++x) nblk += (csz[x] + 15) / 16;
        }
        const size_t lds_tab = (size_t)TFQ1_RPW * N * sizeof(int2) + TFQ1_RPW * sizeof(Q1Row) + ((size_t)TFQ1_RPW * nblk + 4) * sizeof(int);
        const size_t lds_c3 = (size_t)N * n3r * sizeof(double);
        const bool blds = lds_tab + lds_c3 <= (size_t)80 << 10;       // two workgroups per CU
        const int rpw = TFQ1_RPW;
        const unsigned grid = (unsigned)((n_rows + rpw - 1) / rpw);
        const size_t lds = lds_tab + (blds ? lds_c3 : 0);
        if (lds > ((size_t)160 << 10) - 256) { msg = "AO->MO first quarter: the row tables do not fit LDS"; rc = TF_EINVAL; goto done; }
        if (lds > ((size_t)64 << 10)) {
            TFM_HIP(hipFuncSetAttribute((const void *)mo_q1_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            TFM_HIP(hipFuncSetAttribute((const void *)mo_q1_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            TFM_HIP(hipFuncSetAttribute((const void *)mo_q1_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            TFM_HIP(hipFuncSetAttribute((const void *)mo_q1_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        }
        const int nx_cols = (blds && n3 > 16 && n3 <= 20 && !(getenv("TF_Q1_NX") && getenv("TF_Q1_NX")[0] == '0')) ? n3 - 16 : 0;
        if (nx_cols > 0) {
            if (lds > ((size_t)64 << 10)) {
                TFM_HIP(hipFuncSetAttribute((const void *)mo_q1_kernel<1, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                TFM_HIP(hipFuncSetAttribute((const void *)mo_q1_kernel<1, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                TFM_HIP(hipFuncSetAttribute((const void *)mo_q1_kernel<1, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                TFM_HIP(hipFuncSetAttribute((const void *)mo_q1_kernel<1, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            }
            if (nx_cols == 1) hipLaunchKernelGGL((mo_q1_kernel<1, true, 1>), dim3(grid), dim3(TFQ1_THREADS), lds, 0, Q, BL, n3r, rpw, n_rows, nblk);
            else if (nx_cols == 2) hipLaunchKernelGGL((mo_q1_kernel<1, true, 2>), dim3(grid), dim3(TFQ1_THREADS), lds, 0, Q, BL, n3r, rpw, n_rows, nblk);
            else if (nx_cols == 3) hipLaunchKernelGGL((mo_q1_kernel<1, true, 3>), dim3(grid), dim3(TFQ1_THREADS), lds, 0, Q, BL, n3r, rpw, n_rows, nblk);
            else hipLaunchKernelGGL((mo_q1_kernel<1, true, 4>), dim3(grid), dim3(TFQ1_THREADS), lds, 0, Q, BL, n3r, rpw, n_rows, nblk);
        }
        else if (NT == 1 && blds) hipLaunchKernelGGL((mo_q1_kernel<1, true>), dim3(grid), dim3(TFQ1_THREADS), lds, 0, Q, BL, n3r, rpw, n_rows, nblk);
        else if (NT == 1) hipLaunchKernelGGL((mo_q1_kernel<1, false>), dim3(grid), dim3(TFQ1_THREADS), lds, 0, Q, BL, n3r, rpw, n_rows, nblk);
        else if (blds) hipLaunchKernelGGL((mo_q1_kernel<2, true>), dim3(grid), dim3(TFQ1_THREADS), lds, 0, Q, BL, n3r, rpw, n_rows, nblk);
        else hipLaunchKernelGGL((mo_q1_kernel<2, false>), dim3(grid), dim3(TFQ1_THREADS), lds, 0, Q, BL, n3r, rpw, n_rows, nblk);
        TFM_HIP(hipGetLastError());
    }
    // T[p1][nu][y] = sum_mu C1[mu][p1] R[row(mu, nu)][y]: the hand-written bra kernel follows the row map (n1 <= 32)
    {
        const size_t lds1 = (size_t)N * ((n1 + 1) & ~1) * sizeof(double) + (size_t)N * sizeof(int);
        if (n1 > 32 || lds1 > ((size_t)150 << 10)) { msg = "AO->MO transformation: the bra kernel holds at most 32 orbitals in 150 KB of LDS"; rc = TF_EINVAL; goto done; }
        if (lds1 > ((size_t)64 << 10)) {                              // (N > ~450 at 18 orbitals: one workgroup per CU)
            TFM_HIP(hipFuncSetAttribute((const void *)mo_bra1_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
            TFM_HIP(hipFuncSetAttribute((const void *)mo_bra1_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
        }
        const dim3 grid1((unsigned)((Y + TFB1_THREADS - 1) / TFB1_THREADS), (unsigned)N);
        if (n1 <= 16) hipLaunchKernelGGL(mo_bra1_kernel<1>, grid1, dim3(TFB1_THREADS), lds1, 0, dR, dC1, d_rowmap, BL, n1, (int)Y, dT);
        else hipLaunchKernelGGL(mo_bra1_kernel<2>, grid1, dim3(TFB1_THREADS), lds1, 0, dR, dC1, d_rowmap, BL, n1, (int)Y, dT);
        TFM_HIP(hipGetLastError());
    }
    // T2[p1] (Y x n2) = T[p1] (Y x N) * C2^T (N x n2)
    TFM_BLAS(rocblas_dgemm_strided_batched(blas, rocblas_operation_none, rocblas_operation_transpose, (rocblas_int)Y, n2, N, &one, dT, (rocblas_int)Y,
                                           (rocblas_stride)N * Y, dC2, n2, 0, &zero, dT2, (rocblas_int)Y, (rocblas_stride)n2 * Y, n1));
    // out[p1 p2] (n4 x n3) = C4i (n4 x N) * T2[p1 p2]^T (N x n3)
    TFM_BLAS(rocblas_dgemm_strided_batched(blas, rocblas_operation_none, rocblas_operation_transpose, n4, n3, N, &one, dC4i, n4, 0, dT2, n3,
                                           (rocblas_stride)Y, &zero, d_out, n4, (rocblas_stride)n3 * n4, n1 * n2));
    TFM_HIP(hipEventRecord(e1, 0));
    TFM_HIP(hipEventSynchronize(e1));
    if (seconds) { float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1); *seconds = ms * 1e-3; }
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

}  // namespace tfmp2
