"""Engine -- one context of libtunafock per process/GPU: basis set-up, device-resident ERI tensor,
Fock builds and the native RHF cycle.  Thin: every numerical step happens in the HIP library."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ScfOpts, ScfResult, ScfUhfResult, TunaError, f64, i32, ptr
from .molecule import AOList

# SCF convergence thresholds (tuna_util.py:109-116)
SCF_CONVERGENCE = {
    "loose": {"delta_E": 0.000001, "max_DP": 0.00001, "RMS_DP": 0.000001, "commutator": 0.0001, "name": "loose"},
    "medium": {"delta_E": 0.0000001, "max_DP": 0.000001, "RMS_DP": 0.0000001, "commutator": 0.00001, "name": "medium"},
    "tight": {"delta_E": 0.000000001, "max_DP": 0.00000001, "RMS_DP": 0.000000001, "commutator": 0.0000001, "name": "tight"},
    "extreme": {"delta_E": 0.00000000001, "max_DP": 0.0000000001, "RMS_DP": 0.00000000001, "commutator": 0.000000001,
                "name": "extreme"},
}


class Engine:
    def __init__(self, device: int = 0, rank: int = 0, world: int = 1):
        self._L = _lib.lib()
        self._ctx = self._L.tf_create(int(device), int(rank), int(world))
        if not self._ctx:
            raise TunaError(self._L.tf_last_error(None).decode(), -2)
        self.device, self.rank, self.world = device, rank, world
        self.aos: AOList | None = None
        self.n_cart = self.n_sph = self.n_shell = 0
        self.N = 0
        self.spherical = True

    # ---- plumbing --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._L.tf_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            raise TunaError(self._L.tf_last_error(self._ctx).decode(), rc)

    # ---- basis -----------------------------------------------------------------------------
    def set_basis(self, aos: AOList):
        self.aos = aos
        self._keep = (f64(aos.origin), i32(aos.lmn), i32(aos.prim_off), f64(aos.exps), f64(aos.coefs))
        o, l, p, e, c = self._keep
        self._check(self._L.tf_set_basis(self._ctx, aos.n, ptr(o), ptr(l), ptr(p), ptr(e), ptr(c)))
        a, b, s = C.c_int32(), C.c_int32(), C.c_int32()
        self._check(self._L.tf_dims(self._ctx, a, b, s))
        self.n_cart, self.n_sph, self.n_shell = a.value, b.value, s.value
        return self

    def norms(self):
        norm = np.zeros_like(self.aos.exps)
        coef = np.zeros_like(self.aos.exps)
        self._check(self._L.tf_get_norms(self._ctx, ptr(norm), ptr(coef)))
        return norm, coef

    def sph_matrix(self) -> np.ndarray:
        U = np.zeros((self.n_sph, self.n_cart))
        self._check(self._L.tf_get_sph_matrix(self._ctx, ptr(U)))
        return U

    # ---- one-electron ------------------------------------------------------------------------
    def one_electron(self, atom_xyz, atom_charge, dipole_origin, spherical: bool = True):
        n = self.n_sph if spherical else self.n_cart
        xyz, chg, org = f64(np.asarray(atom_xyz).reshape(-1, 3)), f64(atom_charge), f64(dipole_origin)
        S, T, V = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
        D, Q = np.zeros((3, n, n)), np.zeros((3, n, n))
        self._check(self._L.tf_one_electron(self._ctx, len(chg), ptr(xyz), ptr(chg), ptr(org), int(spherical),
                                            ptr(S), ptr(T), ptr(V), ptr(D), ptr(Q)))
        return S, T, V, D, Q

    def cross_overlap(self, other: AOList) -> np.ndarray:
        S = np.zeros((self.n_cart, other.n))
        o, l, p, e, c = f64(other.origin), i32(other.lmn), i32(other.prim_off), f64(other.exps), f64(other.coefs)
        self._check(self._L.tf_cross_overlap(self._ctx, other.n, ptr(o), ptr(l), ptr(p), ptr(e), ptr(c), ptr(S)))
        return S

    # ---- two-electron ------------------------------------------------------------------------
    LAYOUTS = {"auto": -1, "rows": 0, "packed": 1, "tiles": 2}

    def build_eri(self, spherical: bool = True, layout: str | None = None):
        """layout: "packed" (8-fold unique values, the default where the J/K kernel covers N), "rows" ((i >= j) x full [k][l])
        or None / "auto"."""
        if layout is not None:
            self._check(self._L.tf_set_eri_layout(self._ctx, self.LAYOUTS[layout]))
        self._check(self._L.tf_build_eri(self._ctx, int(spherical)))
        self.spherical = spherical
        self.N = self.n_sph if spherical else self.n_cart
        return self

    def eri_storage(self) -> dict:
        b, r = C.c_int64(), C.c_int64()
        n, ld = C.c_int32(), C.c_int32()
        self._check(self._L.tf_eri_storage(self._ctx, b, r, n, ld))
        return {"bytes": b.value, "rows": r.value, "N": n.value, "ld": ld.value,
                "layout": {2: "tiles", 1: "packed"}.get(self._L.tf_eri_layout(self._ctx), "rows")}

    def eri_timings(self) -> dict:
        t = np.zeros(4)
        self._check(self._L.tf_eri_timings(self._ctx, ptr(t)))
        c = np.zeros(3, dtype=np.int64)
        self._check(self._L.tf_eri_counts(self._ctx, ptr(c)))
        f = np.zeros(1)
        self._check(self._L.tf_eri_flops(self._ctx, ptr(f)))
        return {"total_s": t[0], "cart_kernel_s": t[1], "ket_transform_s": t[2], "bra_transform_s": t[3],
                "shell_quartets": int(c[0]), "primitive_shell_quartets": int(c[1]), "component_quartets": int(c[2]),
                "nominal_flops": float(f[0])}

    def copy_eri(self, out: np.ndarray | None = None) -> np.ndarray:
        N = self.N
        if out is None:
            out = np.empty((N, N, N, N))
        assert out.flags.c_contiguous and out.dtype == np.float64 and out.size == N ** 4
        self._check(self._L.tf_copy_eri(self._ctx, ptr(out)))
        return out

    def sample_eri(self, idx) -> np.ndarray:
        idx = i32(idx).reshape(-1, 4)
        out = np.zeros(len(idx))
        self._check(self._L.tf_sample_eri(self._ctx, len(idx), ptr(idx), ptr(out)))
        return out

    def eri_element(self, aos4: AOList) -> float:
        v = C.c_double()
        o, l, p, e, c = f64(aos4.origin), i32(aos4.lmn), i32(aos4.prim_off), f64(aos4.exps), f64(aos4.coefs)
        self._check(self._L.tf_eri_element(self._ctx, ptr(o), ptr(l), ptr(p), ptr(e), ptr(c), C.byref(v)))
        return v.value

    # ---- Fock build --------------------------------------------------------------------------
    def fock_jk(self, P: np.ndarray):
        """J, K for one [N,N] or several [n,N,N] densities (host buffers).  Partial sums when world > 1, unless a communicator is
        attached (comm_init): then the library has summed them over the ranks."""
        P = f64(P)
        if self._L.tf_eri_layout(self._ctx) >= 0 and (P.ndim not in (2, 3) or P.shape[-1] != self.N or P.shape[-2] != self.N):   # (no tensor yet: the library reports that)
            raise ValueError(f"fock_jk: densities must be [{self.N},{self.N}] or [n,{self.N},{self.N}], got {P.shape}")
        nd = 1 if P.ndim == 2 else P.shape[0]
        J, K = np.zeros_like(P), np.zeros_like(P)
        self._check(self._L.tf_fock_jk(self._ctx, nd, ptr(P), ptr(J), ptr(K)))
        return J, K

    def fock_jk_device(self, dP: int, dJ: int, dK: int, n_dens: int = 1, stream: int = 0):
        """Device-pointer variant (integers from tensor.data_ptr()); asynchronous on `stream`."""
        self._check(self._L.tf_fock_jk_device(self._ctx, n_dens, C.c_void_p(dP), C.c_void_p(dJ), C.c_void_p(dK),
                                              C.c_void_p(stream)))

    def jk_profile(self, enable: bool):
        self._check(self._L.tf_jk_profile(self._ctx, int(enable)))

    def jk_profile_read(self):
        """(seconds summed over launches, launches) of the row kernel since profiling was enabled."""
        s, n = C.c_double(), C.c_int64()
        self._check(self._L.tf_jk_profile_read(self._ctx, C.byref(s), C.byref(n)))
        return s.value, n.value

    # ---- SCF -----------------------------------------------------------------------------------
    def orthogonaliser(self, S: np.ndarray):
        S = f64(S)
        n = S.shape[0]
        X, Si = np.zeros((n, n)), np.zeros((n, n))
        sm = C.c_double()
        self._check(self._L.tf_orthogonaliser(self._ctx, n, ptr(S), ptr(X), ptr(Si), C.byref(sm)))
        return X, sm.value, Si

    # ---- Kohn-Sham exchange-correlation ------------------------------------------------------------
    def dft_setup(self, points, weights, functional: str, x_alpha: float = 2 / 3) -> dict:
        """Puts the grid on the device, evaluates the AOs on it and selects the functional (tuna_amd.dft.FUNCTIONALS).
        While set, scf_rhf runs restricted Kohn-Sham; returns {"hfx": HFX_prop, ...} to pass on."""
        from . import dft
        name = functional.upper()
        if name not in dft.FUNCTIONALS:
            raise TunaError(f"Electronic structure method \"{functional}\" is not supported.")
        xn, cn, dfx, hfx, dfc = dft.FUNCTIONALS[name]
        pts = f64(np.asarray(points).reshape(3, -1))
        wts = f64(np.asarray(weights).reshape(-1))
        self._check(self._L.tf_dft_setup(self._ctx, wts.size, ptr(pts), ptr(wts), dft.X_ID[xn], dft.C_ID[cn], dfx, dfc, float(x_alpha)))
        self.functional = {"name": name, "hfx": hfx, "dfx": dfx, "dfc": dfc, "n_points": int(wts.size)}
        return self.functional

    def dft_vxc(self, P):
        """(V_XC, n_electrons_on_grid, E_X * DFX, E_C * DFC) for a closed-shell density (tuna_scf.py:600-654)."""
        import ctypes
        P = f64(P)
        V = np.zeros_like(P)
        n, ex, ec = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(self._L.tf_dft_vxc(self._ctx, ptr(P), ptr(V), ctypes.byref(n), ctypes.byref(ex), ctypes.byref(ec)))
        return V, n.value, ex.value, ec.value

    def dft_clear(self):
        self._check(self._L.tf_dft_clear(self._ctx))
        self.functional = None

    # ---- post-SCF consumers of the resident tensor ------------------------------------------------
    def ao_to_mo(self, C1, C2=None, C3=None, C4=None) -> np.ndarray:
        """(pq|rs) = sum C1[mu,p] C2[nu,q] C3[la,r] C4[si,s] (mu nu|la si); all four default to C1 (tuna_ci.py:204-255)."""
        Cs = [f64(C1)] + [f64(c) if c is not None else None for c in (C2, C3, C4)]
        Cs = [c if c is not None else Cs[0] for c in Cs]
        n = [c.shape[1] for c in Cs]
        out = np.empty(tuple(n))
        self._check(self._L.tf_ao_to_mo(self._ctx, n[0], ptr(Cs[0]), n[1], ptr(Cs[1]), n[2], ptr(Cs[2]), n[3], ptr(Cs[3]), ptr(out)))
        return out

    def mp2_rhf(self, C, eps, n_occ, n_frozen=0) -> dict:
        """RMP2 correlation energy (tuna_mp.py:834-906): {"E_MP2", "E_OS", "E_SS", "seconds"}."""
        Cm, eps = f64(C), f64(eps)
        import ctypes
        os_, ss, t = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self._check(self._L.tf_mp2_rhf(self._ctx, int(n_occ), int(n_frozen), ptr(Cm), ptr(eps), ctypes.byref(os_), ctypes.byref(ss),
                                       ctypes.byref(t)))
        return {"E_MP2": os_.value + ss.value, "E_OS": os_.value, "E_SS": ss.value, "seconds": t.value}

    def diagonalise(self, F, X):
        """(epsilons, molecular_orbitals) = eigh(sym(X^T F X)), C = X C' on the device (scf:222-250)."""
        F, X = f64(F), f64(X)
        n = F.shape[0]
        eps, Cm = np.zeros(n), np.zeros((n, n))
        self._check(self._L.tf_diagonalise(self._ctx, n, ptr(F), ptr(X), ptr(eps), ptr(Cm)))
        return eps, Cm

    def eigh_stats(self):
        """Counters of the eigensolver paths of this engine's workspace (tunafock.h: tf_eigh_stats)."""
        import ctypes as C
        out = (C.c_int64 * 5)()
        self._check(self._L.tf_eigh_stats(self._ctx, out))
        return dict(zip(("refined_solves", "refinement_steps", "refinement_fallbacks", "blocked_solves", "blocked_declined"), (int(v) for v in out)))

    def jk_path_stats(self):
        """Fock builds of the native cycles over the class-diagonal task list / over the full list (tunafock.h: tf_jk_path_stats)."""
        import ctypes as C
        out = (C.c_int64 * 2)()
        self._check(self._L.tf_jk_path_stats(self._ctx, out))
        return {"class_diagonal_passes": int(out[0]), "full_passes_after_test": int(out[1])}

    def _scf_opts(self, *, conv="medium", max_iter=100, diis=True, max_diis=6, damping="dynamic", damping_factor=0.0, max_damping=0.7,
                  hfx=1.0, n_atom_ao=None):
        N = self.N
        conv_d = SCF_CONVERGENCE[conv] if isinstance(conv, str) else conv
        o = ScfOpts()
        o.max_iter, o.use_diis, o.max_diis = max_iter, int(diis), max_diis
        o.damping = {"none": 0, False: 0, None: 0, "dynamic": 1, True: 1, "static": 2}[damping]
        o.damping_factor, o.max_damping = damping_factor, max_damping
        o.conv_delta_E, o.conv_max_DP = conv_d["delta_E"], conv_d["max_DP"]
        o.conv_rms_DP, o.conv_commutator = conv_d["RMS_DP"], conv_d["commutator"]
        o.hfx = hfx
        n_atom_ao = list(n_atom_ao) if n_atom_ao is not None else [N]
        o.n_atoms = len(n_atom_ao)
        o.n_atom_ao[0] = n_atom_ao[0]
        o.n_atom_ao[1] = n_atom_ao[1] if len(n_atom_ao) > 1 else 0
        return o

    def comm_init(self, uid: bytes, comm_rank: int, comm_size: int):
        """tf_comm_init: attaches an RCCL communicator (collective over the ranks that share the tensor); uid = the 128 bytes rank 0 got
        from Engine.comm_unique_id()."""
        buf = C.create_string_buffer(bytes(uid), 128)
        self._check(self._L.tf_comm_init(self._ctx, buf, int(comm_rank), int(comm_size)))
        return self

    @staticmethod
    def comm_unique_id() -> bytes:
        L = _lib.lib()
        buf = C.create_string_buffer(128)
        rc = L.tf_comm_unique_id(buf)
        if rc != 0:
            raise TunaError(L.tf_last_error(None).decode(), rc)
        return buf.raw

    def comm_attached(self) -> bool:
        return bool(self._L.tf_comm_attached(self._ctx))

    def comm_destroy(self):
        self._check(self._L.tf_comm_destroy(self._ctx))

    def set_allreduce(self, hook):
        """tf_set_allreduce: hook(user, device_ptr, count, stream) -> 0 sums `count` doubles at `device_ptr` over the ranks that share
        the tensor (tuna_amd.distributed.attach_allreduce builds it on torch.distributed); None removes it."""
        from ._lib import ALLREDUCE_FN
        self._allreduce_keepalive = ALLREDUCE_FN(hook) if hook is not None else ALLREDUCE_FN()
        self._check(self._L.tf_set_allreduce(self._ctx, self._allreduce_keepalive, None))
        self.has_allreduce = hook is not None

    def scf_rhf(self, S, T, V, P0, E0, n_occ, V_NN, *, X=None, Fext=None, max_iter=100, **opts):
        N = self.N
        o = self._scf_opts(max_iter=max_iter, **opts)
        r = ScfResult()
        P, Cm, F, eps = np.zeros((N, N)), np.zeros((N, N)), np.zeros((N, N)), np.zeros(N)
        table = np.zeros((max_iter, 7))
        r.P, r.C, r.F, r.eps, r.table = (a.ctypes.data for a in (P, Cm, F, eps, table))
        arrs = [f64(S), f64(T), f64(V), None if Fext is None else f64(Fext), None if X is None else f64(X), f64(P0)]
        rc = self._L.tf_scf_rhf(self._ctx, C.byref(o), *[ptr(a) for a in arrs], float(E0), int(n_occ), float(V_NN),
                                C.byref(r))
        res = {"energy": r.energy, "components": np.array(r.components[:]), "n_iter": r.n_iter, "converged": bool(r.converged),
               "P": P, "C": Cm, "F": F, "epsilons": eps, "table": table[:r.n_iter].copy(), "fock_seconds": r.fock_seconds,
               "eig_seconds": r.eig_seconds, "wall_seconds": r.wall_seconds}
        if rc != 0:
            err = TunaError(self._L.tf_last_error(self._ctx).decode(), rc)
            err.partial = res
            raise err
        return res

    def scf_rhf_batch(self, S, T, V, P0s, E0s, n_occ, V_NN, *, X=None, Fexts=None, max_iter=100, **opts):
        """tf_scf_rhf_batch: several restricted cycles on the resident tensor advanced in lockstep inside the library (the finite-field
        evaluations of energy:315-540); the Fock builds of an iteration go through the tensor together.  Returns the result
        dictionaries of tf_scf_rhf in order, each with "rc"; raises if any cycle failed (err.partial = the list)."""
        N, n = self.N, len(P0s)
        o = self._scf_opts(max_iter=max_iter, **opts)
        res_arr = (ScfResult * n)()
        bufs = []
        for k in range(n):
            P, Cm, F, eps, table = np.zeros((N, N)), np.zeros((N, N)), np.zeros((N, N)), np.zeros(N), np.zeros((max_iter, 7))
            res_arr[k].P, res_arr[k].C, res_arr[k].F, res_arr[k].eps, res_arr[k].table = (a.ctypes.data for a in (P, Cm, F, eps, table))
            bufs.append((P, Cm, F, eps, table))
        P0a = [f64(p) for p in P0s]
        Fa = [None if (Fexts is None or f is None) else f64(f) for f in (Fexts if Fexts is not None else [None] * n)]
        p0_ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in P0a])
        f_ptrs = (C.c_void_p * n)(*[(a.ctypes.data if a is not None else None) for a in Fa])
        E0a = f64(np.asarray(E0s, dtype=np.float64))
        rcs = np.zeros(n, dtype=np.int32)
        passes = np.zeros(2, dtype=np.int64)
        arrs = [f64(S), f64(T), f64(V)]
        Xa = None if X is None else f64(X)
        rc = self._L.tf_scf_rhf_batch(self._ctx, n, C.byref(o), ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), f_ptrs, ptr(Xa), p0_ptrs, ptr(E0a),
                                      int(n_occ), float(V_NN), res_arr, ptr(rcs), ptr(passes))
        out = []
        for k in range(n):
            r, (P, Cm, F, eps, table) = res_arr[k], bufs[k]
            out.append({"energy": r.energy, "components": np.array(r.components[:]), "n_iter": r.n_iter, "converged": bool(r.converged),
                        "P": P, "C": Cm, "F": F, "epsilons": eps, "table": table[:r.n_iter].copy(), "fock_seconds": r.fock_seconds,
                        "eig_seconds": r.eig_seconds, "wall_seconds": r.wall_seconds, "rc": int(rcs[k]),
                        "tensor_passes": int(passes[0]), "fock_builds": int(passes[1])})
        if rc != 0:
            err = TunaError(self._L.tf_last_error(self._ctx).decode(), rc)
            err.partial = out
            raise err
        return out

    def scf_uhf(self, S, T, V, P0_alpha, P0_beta, E0, n_alpha, n_beta, V_NN, *, X=None, Fext=None, max_iter=100, **opts):
        """tf_scf_uhf: the unrestricted cycle (scf:1165-1281) on the device; spin quantities come back as pairs (alpha, beta)."""
        N = self.N
        o = self._scf_opts(max_iter=max_iter, **opts)
        r = ScfUhfResult()
        Pt, table = np.zeros((N, N)), np.zeros((max_iter, 7))
        P, Cm, F, eps = np.zeros((2, N, N)), np.zeros((2, N, N)), np.zeros((2, N, N)), np.zeros((2, N))
        r.common.P, r.common.table = Pt.ctypes.data, table.ctypes.data
        for sp in range(2):
            r.P_spin[sp], r.C_spin[sp], r.F_spin[sp], r.eps_spin[sp] = (a[sp].ctypes.data for a in (P, Cm, F, eps))
        arrs = [f64(S), f64(T), f64(V), None if Fext is None else f64(Fext), None if X is None else f64(X), f64(P0_alpha), f64(P0_beta)]
        rc = self._L.tf_scf_uhf(self._ctx, C.byref(o), *[ptr(a) for a in arrs], float(E0), int(n_alpha), int(n_beta), float(V_NN), C.byref(r))
        c = r.common
        res = {"energy": c.energy, "components": np.array(c.components[:]), "n_iter": c.n_iter, "converged": bool(c.converged),
               "P": Pt, "P_spin": P, "C_spin": Cm, "F_spin": F, "epsilons_spin": eps, "table": table[:c.n_iter].copy(),
               "fock_seconds": c.fock_seconds, "eig_seconds": c.eig_seconds, "wall_seconds": c.wall_seconds}
        if rc != 0:
            err = TunaError(self._L.tf_last_error(self._ctx).decode(), rc)
            err.partial = res
            raise err
        return res
