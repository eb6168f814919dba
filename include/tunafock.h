/*
 * tunafock.h -- C ABI of libtunafock.so: the MI355X (gfx950) two-electron-integral / Fock-build /
 * SCF engine that stands behind TUNA's integral module and SCF functions.
 *
 * Every entry point is plain C: pointers and sizes only, row-major float64, no torch/NumPy types.
 * Each one names the reference interface it replaces (h-brough/TUNA v0.12.0; "pyx" =
 * TUNA/tuna_integrals/tuna_integral.pyx, "scf" = TUNA/tuna_scf.py, "kernel" = TUNA/tuna_kernel.py).
 * INTEGRATION.md shows the ctypes stubs a TUNA maintainer would add.
 *
 * Conventions
 *   - return value: 0 on success, negative on failure (TF_E*); tf_last_error() has the message.
 *     No exception crosses this boundary.  There is NO CPU fallback: without a usable GPU every
 *     compute entry point fails with TF_ENODEVICE.
 *   - host arrays are caller-owned; device memory is owned by the context; one context per process
 *     (one process per GPU); calls are synchronous unless they take a stream; a context is
 *     thread-compatible, not thread-safe.
 *   - AO order, geometry (atoms on the z axis), normalisation and the Cartesian->spherical
 *     convention are the reference's.
 */
#ifndef TUNAFOCK_H
#define TUNAFOCK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tf_ctx tf_ctx;

enum {
    TF_OK = 0,
    TF_EINVAL = -1,    /* bad argument / call order                              */
    TF_ENODEVICE = -2, /* no HIP device, or HIP runtime error                    */
    TF_ENOMEM = -3,    /* host or device allocation failed (pyx:1120,1290 MemoryError) */
    TF_ENOTCONV = -4,  /* SCF not converged in max_iter (scf:1435)               */
    TF_ELINALG = -5,   /* rocSOLVER/rocBLAS failure                              */
    TF_EGEOM = -6      /* molecule not on the z axis (kernel:386-388)            */
};

/* ---- life cycle -------------------------------------------------------------------------- */

/* One context per process/GPU.  `device` is the HIP device ordinal; (rank, world) select this
 * process's shard of the (ij) shell-pair rows of the ERI tensor; use
 * rank 0 / world 1 for a single GPU.  Returns NULL on failure (see tf_last_error(NULL)). */
tf_ctx *tf_create(int device, int rank, int world);
void tf_destroy(tf_ctx *ctx);
/* Message of the last failure on `ctx` (or of the last failed tf_create when ctx == NULL). */
const char *tf_last_error(const tf_ctx *ctx);
/* ABI version of the library (major*100 + minor). */
int tf_version(void);

/* Stateless, host only: longest-processing-time assignment of weighted blocks to ranks (a generic
 * helper; replaces the `schedule="dynamic"` load balancing of pyx:1314).  owner[n_blocks]. */
int tf_shard_plan(int n_blocks, const int64_t *weight, int world, int32_t *owner);
/* Stateless, host only: the plan tf_build_eri uses.  Bra shell pairs (A >= B, index A(A+1)/2 + B) of
 * n_shells shells with dim[s] output functions each: for every A the B range is cut into `world`
 * contiguous segments of equal weight (stored values of their rows in `layout`), the heaviest
 * segment going to the least loaded rank -- every rank keeps, for each row index i, long runs of
 * consecutive j, which is what the J/K kernel's 8-row groups want.  owner[n_shells (n_shells+1)/2]. */
int tf_shard_plan_pairs(int n_shells, const int32_t *dim, int layout, int world, int32_t *owner);

/* ---- basis: replaces `Basis.__cinit__` / `Basis.normalize` (pyx:144-210) ------------------ */

/* Stateless helper = Basis.normalize for ONE Cartesian AO: fills norm[nprim] and rescales
 * coefs[nprim] in place, exactly as pyx:174-210 does. */
int tf_normalize(int l, int m, int n, int nprim, const double *exps, double *coefs_inout, double *norm_out);

/* The ordered Cartesian AO list, as `form_basis` (tuna_molecule.py:532-587) hands it to the engine:
 * for AO i: origin[3i..3i+2], lmn[3i..3i+2], primitives prim_off[i]..prim_off[i+1]-1 of exps /
 * coefs_raw (un-normalised contraction coefficients).  The engine normalises (a1), groups
 * consecutive AOs into shells, and builds the shell-pair data (pyx:1050-1128). */
int tf_set_basis(tf_ctx *ctx, int n_ao_cart, const double *origin, const int32_t *lmn, const int32_t *prim_off,
                 const double *exps, const double *coefs_raw);
/* Parity hook for a1: per-primitive norm and normalised coefficients, same layout as exps. */
int tf_get_norms(const tf_ctx *ctx, double *norm, double *coefs_normalised);
/* Dimensions after tf_set_basis: Cartesian / spherical AO counts and number of shells. */
int tf_dims(const tf_ctx *ctx, int *n_cart, int *n_sph, int *n_shell);
/* The Cartesian->spherical matrix U [n_sph, n_cart] (kernel:540-649). */
int tf_get_sph_matrix(const tf_ctx *ctx, double *U);

/* ---- one-electron companions: replace calculate_one_electron_integrals (pyx:282-435) and
 *      calculate_cross_basis_overlap_matrix (pyx:626-768) --------------------------------- */

/* S,T,V [n,n]; D,Q [3,n,n] with n = n_sph if spherical else n_cart (transform kernel:495-502).
 * atom_xyz [3*n_atoms] must be on the z axis; D,Q may be NULL. */
int tf_one_electron(tf_ctx *ctx, int n_atoms, const double *atom_xyz, const double *atom_charge,
                    const double *dipole_origin, int spherical, double *S, double *T, double *V, double *D,
                    double *Q);
/* Overlap between the context's basis (rows) and a second AO list (columns), Cartesian, [n1,n2]. */
int tf_cross_overlap(tf_ctx *ctx, int n_ao2, const double *origin2, const int32_t *lmn2, const int32_t *prim_off2,
                     const double *exps2, const double *coefs_raw2, double *S_cross);

/* ---- two-electron integrals: replace calculate_electron_repulsion_integrals (pyx:1267-1355)
 *      + transform_to_spherical_harmonics (kernel:454-529) -------------------------------- */

/* Build this rank's rows of the (ij|kl) tensor on the device.  spherical = 0 is CARTHARM
 * (kernel:481).  The tensor stays resident in HBM in one of three layouts:
 *   TF_LAYOUT_PACKED (default): the 8-fold unique values without the exact zeros of the x/y parity rule, row (i >= j) = the pairs
 *                               (k >= l) <= (i,j) of its parity class -- ~N^4 / 29 x 8 bytes instead of the reference's 8 N^4 (kernel:349);
 *   TF_LAYOUT_TILES:            the same values in the operand order of v_mfma_f64_16x16x4, second index innermost: the fastest Fock
 *                               pass over FOUR OR MORE densities at once (tf_fock_jk with n_dens >= 4, tf_scf_rhf_batch), slower than
 *                               PACKED for one or two;
 *   TF_LAYOUT_ROWS:             rows (i >= j) x full [k][l] -- 4 N^4 bytes (N > 1024, and the round-1 baseline). */
int tf_build_eri(tf_ctx *ctx, int spherical);
#define TF_LAYOUT_AUTO (-1)
#define TF_LAYOUT_ROWS 0
#define TF_LAYOUT_PACKED 1
#define TF_LAYOUT_TILES 2
/* Layout for the next tf_build_eri (TF_LAYOUT_AUTO = packed when the J/K kernel covers N, i.e. N <= 1024). */
int tf_set_eri_layout(tf_ctx *ctx, int layout);
/* Layout of the stored tensor (TF_LAYOUT_ROWS / TF_LAYOUT_PACKED / TF_LAYOUT_TILES), or TF_EINVAL before tf_build_eri. */
int tf_eri_layout(const tf_ctx *ctx);
/* Alignment unit of the packed layout, in doubles: pair (k >= l) sits at tri(k) + l of its tensor row, where
 * tri(k) = sum over m = 1..k of (m rounded up to the unit); tensor rows are rounded up to the unit as well. */
int tf_packed_pad(void);
/* Bytes of HBM holding the stored rows, number of stored rows, row length (leading dimension). */
int tf_eri_storage(const tf_ctx *ctx, int64_t *bytes, int64_t *n_rows, int32_t *n, int32_t *ld);
/* Dense N^4 tensor with all 8 images, as the reference leaves it in `ERI_AO` (caller-allocated,
 * every element written).  Rows owned by other ranks are filled with zeros when world > 1. */
int tf_copy_eri(tf_ctx *ctx, double *host_out);
/* Values at n_idx index quadruples idx[4*q..4*q+3] = (i,j,k,l) (parity/debug at sizes where the
 * dense copy does not fit the host). */
int tf_sample_eri(tf_ctx *ctx, int64_t n_idx, const int32_t *idx, double *values);
/* One contracted Cartesian integral between four AOs given explicitly
 * (calculate_electron_repulsion_integral, pyx:1376-1414).  Arrays describe 4 AOs like tf_set_basis. */
int tf_eri_element(tf_ctx *ctx, const double *origin, const int32_t *lmn, const int32_t *prim_off,
                   const double *exps, const double *coefs_raw, double *value);

/* ---- Fock build: replaces calculate_coulomb_matrix (scf:55-72) and
 *      calculate_exchange_matrix (scf:27-44) --------------------------------------------- */

/* J_ij = sum_kl (ij|kl) P_kl ;  K_ij = sum_kl (il|kj) P_kl  for n_dens densities, host buffers
 * [n_dens,N,N].  With world > 1 the result is this rank's PARTIAL J and K (sum over ranks =
 * full matrices) and the caller all-reduces -- unless a communicator is attached (tf_comm_init
 * below): then the library has summed them over the ranks already. */
int tf_fock_jk(tf_ctx *ctx, int n_dens, const double *P, double *J, double *K);
/* N > 1 ranks: every rank holds the tensor rows of its bra shell pairs, a Fock build gives partial J and K,
 * and ONE all-reduce of the stacked [J;K] completes them.  The native SCF cycles (tf_scf_rhf / tf_scf_uhf) call this hook once per
 * Fock build on a device buffer of `count` doubles; the caller performs sum-all-reduce over its ranks ordered after / before the work
 * on `stream` (tuna_amd: torch.distributed over RCCL, see tuna_amd/distributed.py).  Returns 0 on success.  Without a registered
 * hook the SCF entry points refuse a sharded tensor. */
typedef int (*tf_allreduce_fn)(void *user, double *device_buf, int64_t count, void *stream);
int tf_set_allreduce(tf_ctx *ctx, tf_allreduce_fn fn, void *user);

/* The exchange step inside the library: an RCCL communicator of the ranks that share the tensor (one process per GPU; RCCL over xGMI).
 * Rank 0 calls tf_comm_unique_id and hands the TF_COMM_ID_BYTES bytes to the other ranks by any means (tuna_amd: torch.distributed's
 * store); then EVERY rank calls tf_comm_init (collective).  From then on the native cycles, tf_ao_to_mo / tf_mp2_rhf and
 * tf_fock_jk_device complete their partial sums with ncclAllReduce on the library's (or the caller's) stream -- no host callback per
 * Fock build; a hook registered with tf_set_allreduce is ignored while a communicator is attached.  librccl is loaded at the first call
 * (dlopen): processes that never shard do not pay for it.  The reference has no counterpart. */
#define TF_COMM_ID_BYTES 128
int tf_comm_unique_id(void *id_out);
int tf_comm_init(tf_ctx *ctx, const void *id, int comm_rank, int comm_size);
int tf_comm_destroy(tf_ctx *ctx);
/* 1 when a communicator is attached: tf_fock_jk_device then returns J and K summed over the ranks. */
int tf_comm_attached(const tf_ctx *ctx);

/* Same with device pointers, asynchronous on `stream` (a hipStream_t, may be NULL for the
 * default stream).  Inputs already in HBM; nothing is synchronised or copied.  With the packed
 * layout the densities must be symmetric here (every SCF density is); tf_fock_jk itself also
 * accepts general matrices (it checks on the host and takes a second pass for them). */
int tf_fock_jk_device(tf_ctx *ctx, int n_dens, const double *dP, double *dJ, double *dK, void *stream);

/* ---- SCF: replaces run_self_consistent_field_cycle (scf:1292-1435) for RHF ---------------- */

typedef struct {
    int32_t max_iter;           /* MAXITER, calc:158 (100)                         */
    int32_t use_diis;           /* DIIS / NODIIS, calc:198,100                     */
    int32_t max_diis;           /* DIIS n (6)                                      */
    int32_t damping;            /* 0 = NODAMP, 1 = dynamic (default), 2 = static   */
    double damping_factor;      /* DAMP x (static)                                 */
    double max_damping;         /* MAXDAMP (0.7), calc:159                         */
    double conv_delta_E;        /* thresholds of util:109-116                      */
    double conv_max_DP;
    double conv_rms_DP;
    double conv_commutator;
    double hfx;                 /* HFX_prop, calc:207 (1.0)                        */
    int32_t n_atom_ao[2];       /* AOs on atom A / B (partition_ranges) for the Mulliken damping */
    int32_t n_atoms;
} tf_scf_opts;

typedef struct {
    double energy;              /* E_total = E_elec + V_NN                         */
    double components[7];       /* kinetic, nuc-el, coulomb, exchange, correlation, field, field-gradient (scf:402) */
    int32_t n_iter;
    int32_t converged;
    double *P;                  /* [N,N] caller-allocated, may be NULL             */
    double *C;                  /* [N,N] molecular orbitals                        */
    double *eps;                /* [N]                                             */
    double *F;                  /* [N,N]                                           */
    double *table;              /* [max_iter,7]: step, E_total, dE, rms(DP), max(DP), commutator, damping (scf:105) */
    double fock_seconds;        /* accumulated device time inside the J/K kernels  */
    double eig_seconds;         /* accumulated time in the eigensolver             */
    double wall_seconds;
} tf_scf_result;

/* Runs the RHF cycle of scf:1072-1154 / 1292-1435 with the tensor built by tf_build_eri.
 * S,T,V,Fext (field terms, may be NULL = 0), X = S^-1/2 (kernel:756-816; NULL => computed here),
 * P0 guess density with guess energy E0, n_occ doubly occupied orbitals. */
int tf_scf_rhf(tf_ctx *ctx, const tf_scf_opts *opts, const double *S, const double *T, const double *V,
               const double *Fext, const double *X, const double *P0, double E0, int n_occ, double V_NN,
               tf_scf_result *out);

/* Unrestricted cycle: run_unrestricted_SCF_cycle (scf:1165-1281) inside the same outer loop.  Both spin densities go
 * through the tensor in one fused pass per iteration.  `common` as for tf_scf_rhf (P = total density; C, eps, F unused);
 * the *_spin arrays are the alpha ([0]) and beta ([1]) quantities, caller-allocated, each may be NULL. */
typedef struct {
    tf_scf_result common;
    double *P_spin[2];          /* [N,N]                                           */
    double *C_spin[2];          /* [N,N]                                           */
    double *eps_spin[2];        /* [N]                                             */
    double *F_spin[2];          /* [N,N]                                           */
} tf_scf_uhf_result;
int tf_scf_uhf(tf_ctx *ctx, const tf_scf_opts *opts, const double *S, const double *T, const double *V,
               const double *Fext, const double *X, const double *P0_alpha, const double *P0_beta, double E0,
               int n_alpha, int n_beta, double V_NN, tf_scf_uhf_result *out);

/* Several restricted cycles on the SAME tensor advanced in lockstep: what the reference's finite-field drivers run one after the other
 * (tuna_energy.py:315-540: 2, 8 or 12 energy evaluations that differ only in the field term F_fld, kernel:660-677).  Arguments as
 * tf_scf_rhf, with one Fext (may be NULL, or entries NULL), P0, E0 and result per cycle; every cycle follows the iteration order of
 * a run on its own (scf:1072-1154) and stops by its own criteria, but the Fock builds of an iteration go through the tensor together
 * (two densities per pass).  rc_out[c] (may be NULL): return code of cycle c; passes_out (may be NULL): {passes over the tensor,
 * Fock builds}.  Returns 0 or the first non-zero cycle code.  Unsharded tensors, Hartree-Fock only. */
int tf_scf_rhf_batch(tf_ctx *ctx, int n_cycles, const tf_scf_opts *opts, const double *S, const double *T, const double *V,
                     const double *const *Fext, const double *X, const double *const *P0, const double *E0, int n_occ, double V_NN,
                     tf_scf_result *out, int32_t *rc_out, int64_t *passes_out);


/* X = S^-1/2, S^-1 and the smallest overlap eigenvalue (kernel:756-816), host buffers [N,N]. */
int tf_orthogonaliser(tf_ctx *ctx, int n, const double *S, double *X, double *S_inv, double *smallest_eig);

/* ---- Kohn-Sham exchange-correlation (BASELINE config 4) ------------------------- */

/* Hands the molecular integration grid (built by the caller exactly as set_up_integration_grid / build_molecular_grid,
 * tuna_dft.py:94-394, do: xyz [3][n_points], weights [n_points]) to the context, which evaluates every AO and its gradient
 * on it once (construct_basis_functions_on_grid, tuna_dft.py:516-666) and keeps them in HBM.  x_functional: 0 none,
 * 1 Slater, 2 B88, 3 B3 (0.9 B88 + 0.1 Slater); c_functional: 0 none, 1 VWN5, 2 VWN3, 3 LYP, 4 3P (0.81 LYP + 0.19 VWN5),
 * 5 3P with VWN3 ("B3LYP/G"); dfx / dfc = DFX_prop / DFC_prop; x_alpha = X_alpha (2/3).  While a grid is set,
 * tf_scf_rhf runs the restricted Kohn-Sham cycle (opts.hfx = HFX_prop).  Needs tf_build_eri first. */
int tf_dft_setup(tf_ctx *ctx, int64_t n_points, const double *xyz, const double *weights, int x_functional, int c_functional, double dfx,
                 double dfc, double x_alpha);
/* V_XC [N,N] = V_X*DFX + V_C*DFC for the closed-shell density P [N,N] (calculate_restricted_exchange_correlation_matrix,
 * tuna_scf.py:600-654), the grid integral of the density and the scaled exchange / correlation energies. */
int tf_dft_vxc(tf_ctx *ctx, const double *P, double *Vxc, double *n_elec, double *e_x, double *e_c);
/* Back to Hartree-Fock. */
int tf_dft_clear(tf_ctx *ctx);

/* ---- consumers of the resident tensor: AO->MO transformation and RMP2 ------------- */

/* out[p,q,r,s] = sum_{mu nu la si} C1[mu,p] C2[nu,q] C3[la,r] C4[si,s] (mu nu|la si)  -- chemists' (pq|rs), host out
 * [n1,n2,n3,n4]; C_k are [N, n_k] row-major host matrices (columns = orbitals).  With all four = the full MO matrix
 * this is transform_ERI_AO_to_MO (tuna_ci.py:204-255); every quarter step is an f64 GEMM (rocBLAS / MFMA).
 * Packed layout: a stored row holds the pairs (la si) <= (mu nu); it is transformed as it lies (one read of every stored value,
 * own rows only) and the other half of the tensor is the transposed result -- two transformations unless (C1, C2) = (C3, C4).
 * world > 1: every rank transforms the rows it owns and the hook of tf_set_allreduce sums the result (all ranks call; without
 * the hook the call is refused, TF_EINVAL). */
int tf_ao_to_mo(tf_ctx *ctx, int n1, const double *C1, int n2, const double *C2, int n3, const double *C3, int n4, const double *C4,
                double *out);
/* Restricted MP2 correlation energy components from canonical orbitals C [N,N], eps [N] (tuna_mp.py:834-906, energy part):
 * *e_os = sum g^2/D, *e_ss = sum g (g - g^T_ab)/D over (ia|jb) with i,j in [n_frozen, n_occ), a,b >= n_occ;
 * E_MP2 = e_os + e_ss.  seconds (may be NULL): wall time of transform + energy.  world > 1: as tf_ao_to_mo (one all-reduce of
 * the (ia|jb) block, then every rank evaluates the same sums). */
int tf_mp2_rhf(tf_ctx *ctx, int n_occ, int n_frozen, const double *C, const double *eps, double *e_os, double *e_ss, double *seconds);

/* eps[N], C[N,N] = eigenpairs of the Fock matrix in the orthogonalised basis, C = X C' (diagonalise_Fock_matrix,
 * scf:222-250): rocBLAS dgemm + rocSOLVER dsyevd; host buffers. */
int tf_diagonalise(tf_ctx *ctx, int n, const double *F, const double *X, double *eps, double *C);

/* ---- instrumentation ------------------------------------------------------------------- */

/* Device-side seconds of the last tf_build_eri broken down by stage:
 * [0] total, [1] primitive/contracted Cartesian kernel, [2] ket transform, [3] bra transform+store. */
int tf_eri_timings(const tf_ctx *ctx, double *seconds4);
/* Work counters of the last tf_build_eri: [0] shell quartets, [1] primitive shell quartets,
 * [2] Cartesian component quartets. */
int tf_eri_counts(const tf_ctx *ctx, int64_t *counts3);
/* The reference algorithm's floating-point operation count for the quartets the last tf_build_eri evaluated (SURVEY.md section 8d(ii):
 * per primitive AO quartet that passes the parity test pyx:1324-1327, 8 x the inner terms of the loop nest pyx:1179-1217 plus the
 * Boys / R table cost 6 (L+1) + 3 (L+1)^2 / 2 + 60).  The device kernels factorise the sums and execute fewer operations; the
 * figure prices the build against the FP64 vector peak. */
int tf_eri_flops(const tf_ctx *ctx, double *nominal_flops);
/* Alignment unit (in doubles) of the stored segments of the packed, parity-blocked tensor layout. */
int tf_segment_pad(void);

/* Seconds per symmetric eigensolve (random n x n matrix) of the solver variants considered for a12
 * (0 = rocsolver dsyevd, 1 = dsyev, 2 = dsyevj); instrumentation only. */
int tf_eigh_probe(tf_ctx *ctx, int n, int variant, int reps, double *seconds);

/* Counters of the context's eigensolver paths since tf_create (instrumentation, tests): out[0] solves by eigenvector refinement,
 * out[1] refinement steps, out[2] refinements that fell back to a full eigensolve, out[3] eigensolves done block by block
 * (the x/y parity classes of a diatomic solved together, tf_scf.hip.h: eigh_blocked), out[4] eigensolves where the matrix did not
 * have the block structure and the full matrix was solved. */
int tf_eigh_stats(tf_ctx *ctx, int64_t out[5]);

/* Fock builds of the native cycles on the packed layout since tf_create: out[0] builds that went over the shorter task list of a
 * class-diagonal density (no element between AOs of different x/y parity: every product of the skipped tasks is an exact zero,
 * J and K are bit for bit those of the full list), out[1] builds whose density was not and took the full list.  Builds through
 * tf_fock_jk / tf_fock_jk_device always take the full list (no test, no read-back). */
int tf_jk_path_stats(tf_ctx *ctx, int64_t out[2]);

/* HIP-event timing of the dominant kernel of the Fock build (the row pass over the stored tensor), recorded on
 * the stream each build is launched on.  enable: start (and reset) / stop collecting; read: synchronises the
 * recorded events, returns their summed duration and the number of launches, and resets. */
int tf_jk_profile(tf_ctx *ctx, int enable);
int tf_jk_profile_read(tf_ctx *ctx, double *seconds_total, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* TUNAFOCK_H */
