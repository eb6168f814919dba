// tf_layout.hip.h -- device view of the parity-blocked tensor layout (described at the top of tf_jkpacked.hip.h; NumPy model:
// tests/layout_model.py).  Shared by the ERI generation kernels (which write slab rows in the complete-row shape), the slab
// transforms and the Fock-build kernels.
#pragma once
#include <hip/hip_runtime.h>

#ifndef TF_SEG_PAD
#define TF_SEG_PAD 8             // segments start at multiples of this many doubles (even; 8 = 64 bytes, 16 = one 128-byte line)
#endif

// ---- device view of the layout tables (built by tf_build_eri; every index "internal" unless it says original) ---------------
struct KInfo { int offA, cnt; };  // segment of AO k in a row of class c: offset inside the section of k's class, stored values
// The small per-class tables live in device memory (itab / ltab), not in the by-value struct: kernels index them with run-time
// class numbers, and a dynamically indexed kernel argument would be copied to scratch memory.
enum { BL_CSTART = 0, BL_CSIZE = 4, BL_WFIRST = 8, BL_FULLSEC = 13, BL_GBASE = 29, BL_ITAB = 33 };   // itab offsets
enum { BL_CBASE = 0, BL_NP = 4, BL_LTAB = 8 };                                                         // ltab offsets
struct BLayout {
    int N, NW, RS;                // AOs; column chunks; doubles of a row's "row part" vector
    long long NPtot;              // pair index space of all classes
    const int *itab;              // cstart[4], csize[4] (internal range of each class), wfirst[5] (chunks of class b: wfirst[b] ..
                                  // wfirst[b + 1]), fullsec[4][4] ([c][a]: start of section a in a complete class-c row), gbase[4]
                                  // (granule table of class c starts at gk[gbase[c]])
    const long long *ltab;        // cbase[4], NP[4]: pair index space of class c is [cbase[c], cbase[c] + NP[c])
    const int *ao;                // [N] by ORIGINAL index: class | loc << 2
    const int *origI;             // [N] internal -> original
    const int *clsI;              // [N] class by internal index
    const KInfo *kinfo;           // [4][N]
    const int *cntA;              // [4][N]: AOs of class a with original index <= that of internal x  (walk limit of bra index x)
    const int *kap0;              // [4][NW]: first member of class b(w) ^ c whose segment reaches chunk w
    const int *kapF;              // [4][NW]: first member from which all 128 columns of chunk w are strictly below k
    const int *rpoff;             // [4][NW]: row parts of chunk w inside a row part vector
    const int *chunk_c0, *chunk_width, *chunk_cls;   // [NW]
    const int *chunk_of;          // [N] chunk of an internal column
    const int *gk;                // per class: AO k of the segment that holds granule g (TF_SEG_PAD doubles) of the pair index space
};
__device__ __forceinline__ int bl_cstart(const BLayout &L, int a) { return L.itab[BL_CSTART + a]; }
__device__ __forceinline__ int bl_wfirst(const BLayout &L, int b) { return L.itab[BL_WFIRST + b]; }
__device__ __forceinline__ int bl_fullsec(const BLayout &L, int c, int a) { return L.itab[BL_FULLSEC + 4 * c + a]; }
__device__ __forceinline__ int bl_gbase(const BLayout &L, int c) { return L.itab[BL_GBASE + c]; }
__device__ __forceinline__ long long bl_cbase(const BLayout &L, int c) { return L.ltab[BL_CBASE + c]; }
__device__ __forceinline__ long long bl_np(const BLayout &L, int c) { return L.ltab[BL_NP + c]; }
__device__ __forceinline__ int ao_cls(int w) { return w & 3; }
__device__ __forceinline__ int ao_loc(int w) { return w >> 2; }
__device__ __forceinline__ int ao_sigma(const BLayout &L, int w) { return bl_cstart(L, w & 3) + (w >> 2); }

