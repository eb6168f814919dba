// Microbenchmark: HBM read rate of the access patterns the Fock kernel could use (tools/README.md).
// Memory = U units of [K steps][8 rows][W columns] doubles (W = 256: four 64-column chunks), as a storage unit of the packed tensor.
//   stream : every wave reads its own contiguous 8 KB per step (the ceiling)
//   pieces : a wave owns (two units, one chunk): per step 16 pieces of 512 B (8 rows x 2 units); the four waves of a workgroup work on
//            four different unit pairs; chunk-major task order (the other chunks of a unit are read much later)
//   pieces_near : the same, chunk-minor task order (the four chunks of a unit pair are four consecutive workgroups)
//   rows   : a workgroup owns (two units), wave w reads chunk w: the workgroup reads contiguous 2 x 16 KB per step
//   pieces256 / pieces1k : four units per wave on 32 columns each (256-byte pieces) / one unit per wave on 128 columns (1 KB pieces)
//   masked : as pieces, through buffer loads with the lanes beyond a per-step limit out of range (the staircase of the packed rows:
//            the limit grows from 0 to the chunk width over the steps, half of the lanes load nothing) -- is a load instruction
//            with idle lanes cheaper?
// build: hipcc -O3 --offload-arch=gfx950 hbm_patterns.hip -o hbm_patterns ; run: ./hbm_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

typedef double v2d __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int K = 75, R = 8, W = 256, CW = 64;
constexpr long long UNIT = (long long)K * R * W;      // doubles

__device__ __forceinline__ v2d ldg(const double *p) { return __builtin_nontemporal_load((const v2d *)p); }

__device__ __forceinline__ v2d buf_ld(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned soff)
{
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, (int)soff, 2);
    return __builtin_bit_cast(v2d, v);
}

// mode 0: stream; 1: pieces (chunk-major); 2: pieces (chunk-minor); 3: rows
template <int MODE>
__global__ __launch_bounds__(256, 2) void read_kernel(const double *__restrict__ T, long long units, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int h = lane >> 5, q = lane & 31;
    const long long npairs = units / 2;
    double acc = 0.0;
    if (MODE == 5 || MODE == 6) {
        // 5: four units per wave, 16 lanes x 16 B = 256-byte pieces; 6: one unit per wave, 64 lanes x 16 B = 1 KB pieces
        constexpr int G = MODE == 5 ? 4 : 1, LG = 64 / G, NCH = W / (2 * LG);          // groups per wave, lanes per group, chunks per unit
        const long long t = (long long)blockIdx.x * 4 + w;                              // units / G * NCH tasks
        const long long ngrp = units / G;
        const int chunk = (int)(t / ngrp); const long long grp = t % ngrp;
        if (chunk < NCH) {
        const double *base = T + (G * grp + lane / LG) * UNIT + chunk * (2 * LG) + 2 * (lane % LG);
        v2d a[2][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[0][r] = ldg(base + r * W);
        for (int s = 0; s < K; ++s) {
            const double *nx = base + (long long)min(s + 1, K - 1) * (R * W);
#pragma unroll
            for (int r = 0; r < 8; ++r) a[(s + 1) & 1][r] = ldg(nx + r * W);
#pragma unroll
            for (int r = 0; r < 8; ++r) acc += a[s & 1][r].x + a[s & 1][r].y;
        }
        }
    } else if (MODE == 4) {
        const long long t = (long long)blockIdx.x * 4 + w;
        const int chunk = (int)(t / npairs); const long long pair = t % npairs;
        const double *base = T + 2 * pair * UNIT + chunk * CW;
        const unsigned d1 = 8u * (unsigned)UNIT;
        v2d a[2][8];
        auto issue = [&](int s, v2d (&dst)[8]) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(base + (long long)s * (R * W)), 0, 0x7fffffff, 0x00020000);
            const int lim = (s * 32 + K - 1) / K;                          // lanes q < lim hold a value: 0 .. 32 over the steps
            const unsigned off = q < lim ? 16u * q + (h ? d1 : 0u) : 0x80000000u;
#pragma unroll
            for (int r = 0; r < 8; ++r) dst[r] = buf_ld(rs, off, 8u * (unsigned)(r * W));
        };
        issue(0, a[0]);
        for (int s = 0; s < K; s += 2) {
            issue(min(s + 1, K - 1), a[1]);
#pragma unroll
            for (int r = 0; r < 8; ++r) acc += a[0][r].x + a[0][r].y;
            issue(min(s + 2, K - 1), a[0]);
#pragma unroll
            for (int r = 0; r < 8; ++r) acc += a[1][r].x + a[1][r].y;
        }
    } else if (MODE == 0) {
        // wave-contiguous: task = blockIdx * 4 + w reads K * 4 steps of 8 KB... same bytes per task as the others: one unit pair quarter
        const long long task = (long long)blockIdx.x * 4 + w;            // npairs * 4 tasks
        const double *base = T + task * (UNIT / 2);                       // half a unit pair = 2 units / 4
        v2d a[2][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[0][r] = ldg(base + r * 128 + 2 * lane);
        for (int s = 0; s < K; ++s) {
            const double *nx = base + (long long)min(s + 1, K - 1) * 1024;
#pragma unroll
            for (int r = 0; r < 8; ++r) a[(s + 1) & 1][r] = ldg(nx + r * 128 + 2 * lane);
#pragma unroll
            for (int r = 0; r < 8; ++r) acc += a[s & 1][r].x + a[s & 1][r].y;
        }
    } else {
        long long pair; int chunk;
        if (MODE == 1) { const long long t = (long long)blockIdx.x * 4 + w; chunk = (int)(t / npairs); pair = t % npairs; }
        else if (MODE == 2) { chunk = blockIdx.x & 3; pair = (long long)(blockIdx.x >> 2) * 4 + w; }
        else { chunk = w; pair = blockIdx.x; }
        if (pair >= npairs) return;
        const double *base = T + (2 * pair + h) * UNIT + chunk * CW + 2 * q;
        v2d a[2][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) a[0][r] = ldg(base + r * W);
        for (int s = 0; s < K; ++s) {
            const double *nx = base + (long long)min(s + 1, K - 1) * (R * W);
#pragma unroll
            for (int r = 0; r < 8; ++r) a[(s + 1) & 1][r] = ldg(nx + r * W);
#pragma unroll
            for (int r = 0; r < 8; ++r) acc += a[s & 1][r].x + a[s & 1][r].y;
        }
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

int main(int argc, char **argv)
{
    const double GB = argc > 1 ? atof(argv[1]) : 7.0;
    long long units = (long long)(GB * 1e9 / (UNIT * 8));
    units &= ~7LL;
    const size_t bytes = (size_t)units * UNIT * 8;
    double *T, *out;
    CHECK(hipMalloc((void **)&T, bytes));
    CHECK(hipMalloc((void **)&out, 8));
    CHECK(hipMemset(T, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long long npairs = units / 2;
    const char *names[7] = {"stream", "pieces", "pieces_near", "rows", "masked", "pieces256", "pieces1k"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 7; ++mode) {
            const unsigned grid = (unsigned)npairs;      // every mode: npairs workgroups of 4 waves, a wave reads K steps x 8 KB
            float best = 1e30f;
            for (int it = 0; it < 5; ++it) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(read_kernel<0>, dim3(grid), dim3(256), 0, 0, T, units, out);
                if (mode == 1) hipLaunchKernelGGL(read_kernel<1>, dim3(grid), dim3(256), 0, 0, T, units, out);
                if (mode == 2) hipLaunchKernelGGL(read_kernel<2>, dim3(grid), dim3(256), 0, 0, T, units, out);
                if (mode == 3) hipLaunchKernelGGL(read_kernel<3>, dim3(grid), dim3(256), 0, 0, T, units, out);
                if (mode == 4) hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, T, units, out);
                if (mode == 5) hipLaunchKernelGGL(read_kernel<5>, dim3(grid), dim3(256), 0, 0, T, units, out);
                if (mode == 6) hipLaunchKernelGGL(read_kernel<6>, dim3(grid), dim3(256), 0, 0, T, units, out);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double moved = mode == 4 ? bytes * 0.5 : (double)bytes;   // masked: about half of the lanes load
            printf("%-12s %.3f ms  %.2f TB/s  (%.2f GB)\n", names[mode], best, moved / (best * 1e-3) / 1e12, moved / 1e9);
        }
    return 0;
}
