#!/usr/bin/env python3
"""bench.py -- Fock builds/s (+ SCF wall time) of the MI355X engine, with roofline and CPU baseline.

    python bench.py --gpus N --steps K --warmup W [--workload synth-400|n2-cc-pvtz|ar2-cc-pvqz|synth-<N>]

A "step" is ONE Fock build: J and K (scf:70, scf:42 of the reference) for one density matrix from the ERI tensor
resident in HBM, then -- for N > 1 ranks -- one RCCL all-reduce of the stacked [J;K] (each rank holds a shard of
the (ij) shell-pair rows of the tensor).  Inputs (tensor and density) are resident in HBM before the timed region.
One JSON line on rank 0; see DESIGN.md section "Measurement" for every field.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # before torch initialises HIP (tuna_amd/__init__.py: concurrent launches of the tensor build)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
FP64_MATRIX_PEAK_FLOPS = 78.6e12 # MI355X FP64 matrix (MFMA) peak: the FP64 matrix cores run at the FP64 vector rate on CDNA4
FP64_VECTOR_PEAK_FLOPS = 78.6e12 # MI355X FP64 vector peak (half the 157.3 TFLOP/s FP32 vector figure of MI355X_MICROARCH.md)
ANCHOR_N2_CCPVTZ = -108.9834703056   # SURVEY.md section 6.2 (reference engine + reference tuna_scf.py, golden/c2)


def build_workload(name: str):
    from tuna_amd import molecule as mol
    name = name.lower()
    if name.startswith("synth-"):
        n = int(name.split("-")[1])
        counts = mol.synthetic_counts(n)
        atoms = mol.make_atoms(["AR", "AR"], 7.1)
        shells = mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)})
        desc = (f"synthetic even-tempered uncontracted Ar2-like diatomic, {counts[0]}s{counts[1]}p{counts[2]}d{counts[3]}f per atom, "
                f"R = 7.1 a0 (SURVEY.md section 8d)")
        nocc = 18
    elif name == "n2-cc-pvtz":
        atoms = mol.make_atoms(["N", "N"], mol.angstrom_to_bohr(1.0977))
        shells = mol.build_shells(atoms, "cc-pVTZ")
        desc, nocc = "N2 RHF/cc-pVTZ, R = 1.0977 A (BASELINE.json configs[1])", 7
    elif name == "ar2-cc-pvqz":
        atoms = mol.make_atoms(["AR", "AR"], mol.angstrom_to_bohr(3.76))
        shells = mol.build_shells(atoms, "cc-pVQZ")
        desc, nocc = "Ar2 RHF/cc-pVQZ, R = 3.76 A (BASELINE.json configs[2])", 18
    else:
        raise SystemExit(f"unknown workload {name}")
    return atoms, shells, mol.expand_cartesian_aos(shells), nocc, desc


def _time_fock_einsums(Ns: int, budget_s: float, max_n: int = 200, warm: bool = True):
    """seconds per J+K build with the reference's einsum strings on a dense Ns^4 tensor of random values"""
    from oracle import scf_oracle as so
    rng = np.random.default_rng(0)
    if Ns <= 128:
        T = rng.standard_normal((Ns, Ns, Ns, Ns))
    else:                                       # (GBs of normal deviates cost more than the builds: one random slab, rescaled per first index)
        slab = rng.standard_normal((Ns, Ns, Ns))
        T = np.empty((Ns, Ns, Ns, Ns))
        for i in range(Ns):
            np.multiply(slab, 1.0 + 0.37 * np.sin(i), out=T[i])
    A = rng.standard_normal((Ns, Ns))
    P = A + A.T
    if warm:
        so.coulomb(P, T); so.exchange(P, T)
    t0 = time.perf_counter()
    n = 0
    while True:
        so.coulomb(P, T)
        so.exchange(P, T)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= max_n:
            break
    return (time.perf_counter() - t0) / n, n


def cpu_baseline_fock(N_workload: int, budget_s: float = 9.0):
    """The reference's CPU Fock build -- np.einsum("ijkl,kl->ij") + np.einsum("ilkj,kl->ij"), optimize=True
    (scf:70, scf:42, restated in oracle/scf_oracle.py) -- timed DIRECTLY on dense tensors of two real sizes: N = 118 (Ar2/cc-pVQZ,
    BASELINE.json configs[2]; 1.6 GB) and N = 200 (12.8 GB, two builds), with all the CPUs this process may use; `value` is the
    N = 200 time scaled by (N/200)^4 to the workload (a factor 16 at N = 400, whose dense tensor -- 205 GB -- no host einsum can hold
    twice).  A 96^4 sample with many builds and TUNA's default of 4 threads (tuna_calc.py:153) are reported beside it."""
    import tuna_amd
    # BLAS threads = the CPUs this process may use (cgroup quota): more threads than that only spin and get the process throttled
    threads = tuna_amd.cpu_quota()
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    Ns = min(N_workload, 96)
    sizes = [n for n in (118, 200) if n <= N_workload] or [N_workload]
    pool = threadpool_limits(limits=threads) if threadpool_limits else None
    per_build, n = _time_fock_einsums(Ns, 0.5 * budget_s)
    direct = []
    for nd_ in sizes:
        pb, nb = _time_fock_einsums(nd_, budget_s if nd_ <= 128 else 2.0 * budget_s, 12 if nd_ <= 128 else 2, warm=nd_ <= 128)
        direct.append({"n": nd_, "builds": nb, "ms_per_build": pb * 1e3, "dense_tensor_GB": 8.0 * nd_ ** 4 / 1e9,
                       "scaled_seconds_per_build_at_workload_size": pb * (N_workload / nd_) ** 4,
                       "ratio_to_96_sample": pb * (N_workload / nd_) ** 4 / (per_build * (N_workload / Ns) ** 4)})
    if pool is not None:
        pool.restore_original_limits()
    pool = threadpool_limits(limits=min(4, threads)) if threadpool_limits else None
    per_build4, n4 = _time_fock_einsums(Ns, 0.4 * budget_s, 60)
    if pool is not None:
        pool.restore_original_limits()
        tuna_amd.limit_host_threads()
    big = direct[-1]
    scaled = big["scaled_seconds_per_build_at_workload_size"]
    return {"value": 1.0 / scaled, "unit": "Fock builds/s", "cores": int(threads), "kind": "port",
            "sample": f"{big['builds']} J+K builds with the reference einsum strings on a dense {big['n']}^4 f64 tensor of random values "
                      f"({big['ms_per_build']:.0f} ms each, timed directly), scaled by (N/{big['n']})^4 to N = {N_workload}",
            "seconds_per_build_at_workload_size": scaled,
            "direct": direct,
            "sample_96": {"n": Ns, "builds": n, "ms_per_build": per_build * 1e3,
                          "scaled_seconds_per_build_at_workload_size": per_build * (N_workload / Ns) ** 4,
                          "note": "ratio_to_96_sample above 1 in `direct` = the strided exchange einsum loses cache locality as N grows: "
                                  "an N^4 extrapolation from a small sample flatters the CPU (rounds 1-3 quoted that one)"},
            "tuna_default_4_threads": {"threads": int(min(4, threads)), "builds": n4, "ms_per_build_sample": per_build4 * 1e3,
                                       "value": 1.0 / (per_build4 * (N_workload / Ns) ** 4), "unit": "Fock builds/s"}}


def cpu_baseline_eri(aos, limit_s: float = 20.0):
    """ERI build on the host for the SCF leg: the compiled reference engine when oracle/_ref is present, else the C port -- with all
    the CPUs of the process and with one thread (the reference's OpenMP loop scales very differently from host to host: in the
    8-vCPU build container of SURVEY.md section 6.2 N2/cc-pVTZ takes 1.8 s on 1 thread, 0.7 s on 4 and 1.25-1.44 s on 8 -- oversubscribed
    vCPUs -- while a GPU box's 16 dedicated cores run it in well under 0.1 s)."""
    from oracle import oracle as orc
    if aos.n > 80:
        return None
    kind = "reference" if orc.ref_engine() is not None else "port"
    import tuna_amd
    cores = tuna_amd.cpu_quota()
    fn = orc.ref_eri if kind == "reference" else orc.eri
    fn(aos, cores)                                               # (first call: library load, page faults of the output tensor)
    t0 = time.perf_counter()
    fn(aos, cores)
    t_all = time.perf_counter() - t0
    t0 = time.perf_counter()
    fn(aos, 1)
    t_one = time.perf_counter() - t0
    return {"seconds": t_all, "kind": kind, "cores": cores, "seconds_1_thread": t_one, "speedup_over_1_thread": t_one / t_all}


JK_KERNEL = {"packed": "jk_packed_kernel", "rows": "tfk::jk_rows_kernel", "tiles": "jk_tile_kernel"}


def pmc_traffic(workload: str, world: int, layout: str, stored_bytes: float):
    """HBM bytes per BUILD of the J/K kernel from the committed rocprofv3 PMC passes (profiles/), corrected as
    MI355X_MICROARCH.md (section HBM) prescribes for gfx950: FETCH_SIZE (KB) x 1024 x 2 -- the counter tallies the 128-byte requests
    of 16-byte-per-lane streaming reads at 64 bytes -- plus WRITE_SIZE (KB) x 1024 (exact for 16-byte-per-lane streaming stores).
    Counters cannot be collected from inside an un-profiled run, so this is the figure of the profiled run of the same workload.
    The profile carries `_meta` (commit and stored bytes of the run it was taken from); a profile whose stored bytes differ from this
    run's (another layout or padding) is not used.  Returns (bytes, file, meta) or (None, None, None)."""
    if world != 1:
        return None, None, None
    for rnd in ("r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{workload.replace('-', '')}_{layout}.json")
        try:
            d = json.load(open(path))
            meta = d.get("_meta", {})
            if meta.get("stored_bytes") not in (None, stored_bytes):
                continue
            k = [v for name, v in d.items() if JK_KERNEL[layout] in name][0]
            # a build takes several launches of the kernel (one per workgroup size: 4, 2, 1 waves): bytes per BUILD = the sum over the
            # dispatches of the profiled run / its builds (jk_reduce_kernel runs once per build)
            red = [v for name, v in d.items() if "jk_reduce_kernel" in name]
            per_build = lambda c: (k[c]["sum"] / red[0][c]["dispatches"]) if red and red[0][c]["dispatches"] else k[c]["mean_KB"]
            return (per_build("FETCH_SIZE") * 1024.0 * 2.0 + per_build("WRITE_SIZE") * 1024.0), os.path.relpath(path, ROOT), meta
        except Exception:
            continue
    return None, None, None


def pmc_eri_valu(workload: str, cart_kernel_s: float):
    """EXECUTED vector work of the Cartesian ERI kernels (eri_*; the slab transforms are excluded) from the committed rocprofv3 PMC pass of
    a tensor build of the same workload: wave-level VALU instructions per build (SQ_INSTS_VALU) and the share of the chip's VALU issue
    slots they fill during this run's ERI kernel time -- a wave64 instruction occupies its SIMD's 16-lane FP64 pipe for 4 cycles, 1024
    SIMDs at 2.4 GHz (MI355X_MICROARCH.md).  Unlike `nominal_flops` this counts what the kernels execute, not the reference's loop nest."""
    for rnd in ("r04", "r03"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_eri_{workload.replace('-', '')}.json")
        try:
            d = json.load(open(path))
            builds = float(d["_meta"]["builds"])
            fam = {k: v for k, v in d.items() if k.startswith("eri_")}
            insts = sum(v["SQ_INSTS_VALU"] for v in fam.values()) / builds
            busy = sum(v.get("SQ_BUSY_CYCLES", 0.0) for v in fam.values())
            active = sum(v.get("SQ_ACTIVE_INST_VALU", 0.0) for v in fam.values())
            return {"valu_wave_instructions_per_build": insts,
                    "valu_issue_utilisation": insts * 4.0 / (1024 * 2.4e9 * max(cart_kernel_s, 1e-12)),
                    "lds_wave_instructions_per_build": sum(v["SQ_INSTS_LDS"] for v in fam.values()) / builds,
                    "salu_wave_instructions_per_build": sum(v["SQ_INSTS_SALU"] for v in fam.values()) / builds,
                    "profile": os.path.relpath(path, ROOT), "profile_meta": d["_meta"],
                    "note": "valu_issue_utilisation = VALU wave instructions of the profiled build x 4 cycles / (1024 SIMDs x 2.4 GHz x the ERI "
                            "kernel seconds of THIS run); the kernels run concurrently on 8 streams, so kernel seconds are the device-busy span"}
        except Exception:
            continue
    return None


def pmc_mfma(workload: str, nd: int, kernel_avg_s: float):
    """Matrix-core use of the tiles layout's wide Fock pass (densities = B-operand columns of v_mfma_f64_16x16x4_f64) from the committed
    PMC pass of the same workload: MFMA instructions per launch, the share of the chip's SIMD cycles the matrix pipes are busy during this
    run's kernel time (SQ_VALU_MFMA_BUSY_CYCLES = 64 cycles per instruction; 1024 SIMDs, 2.4 GHz), and the share of those that works on
    real columns (densities per pass / 16)."""
    path = os.path.join(ROOT, "profiles", f"r04_pmc_{workload.replace('-', '')}_tiles_wide.json")
    try:
        d = json.load(open(path))
        per_pass = min(nd, 8)
        key = "jk_tile_kernel<8, 1, 2>" if per_pass > 4 else "jk_tile_kernel<4, 1, 2>"
        k = [v for name, v in d.items() if key in name][0]
        val = lambda c: k[c]["mean"] if isinstance(k[c], dict) else k[c]
        busy = val("SQ_VALU_MFMA_BUSY_CYCLES") / (max(kernel_avg_s, 1e-12) * 2.4e9 * 1024)
        return {"kernel": key, "mfma_instructions_per_launch": val("SQ_INSTS_MFMA"), "cycles_per_instruction": val("SQ_VALU_MFMA_BUSY_CYCLES") / val("SQ_INSTS_MFMA"),
                "matrix_pipe_busy_fraction": busy, "useful_column_fraction": per_pass / 16.0, "useful_busy_fraction": busy * per_pass / 16.0,
                "valu_wave_instructions_per_launch": val("SQ_INSTS_VALU"), "profile": os.path.relpath(path, ROOT), "profile_meta": d.get("_meta")}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 100 timed builds after 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=os.environ.get("TUNA_BENCH_WORKLOAD", "synth-400"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scf", action="store_true")
    ap.add_argument("--n-dens", type=int, choices=[1, 2, 4, 8, 16], default=1,
                    help="densities per step (2 = a UHF build: alpha and beta in one pass; 4, 8, 16 = a finite-field batch: pairs of densities per pass)")
    ap.add_argument("--layout", choices=["packed", "rows", "tiles"], default="packed", help="ERI storage layout (tunafock.h: tf_set_eri_layout)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: tuna_amd has no CPU fallback")
    # TUNA_BENCH_BACKEND=gloo: rehearsal of the multi-rank flow on a box with fewer GPUs than ranks (RCCL refuses two ranks on one
    # device): the ranks share the cards round-robin and the all-reduce of [J;K] is staged through the host.  Not a measurement.
    backend = os.environ.get("TUNA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from tuna_amd.engine import Engine
    atoms, shells, aos, nocc, desc = build_workload(args.workload)
    eng = Engine(local_rank, rank, world)
    eng.set_basis(aos)
    t0 = time.perf_counter()
    eng.build_eri(True, layout=args.layout)
    torch.cuda.synchronize()
    eri_wall_cold = time.perf_counter() - t0                   # includes the one-time allocation of the tensor and slab buffers
    t0 = time.perf_counter()
    eng.build_eri(True, layout=args.layout)                    # what a geometry step pays: the buffers are kept by the context
    torch.cuda.synchronize()
    eri_wall = time.perf_counter() - t0
    N = eng.N
    st = eng.eri_storage()
    eri_t = eng.eri_timings()

    # density: P = A + A^T, A ~ N(0,1) (default_rng(0)), scaled to tr(PS) = n_elec
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    S, T, V, _, _ = eng.one_electron(xyz, chg, [0, 0, 0.5 * atoms[-1].origin[2]])
    A = np.random.default_rng(0).standard_normal((N, N))
    P = A + A.T
    P *= 2 * nocc / np.trace(P @ S)
    nd = args.n_dens
    if nd <= 2:
        dens = [P, 0.5 * P + 0.25 * np.diag(np.diag(P))][:nd]
    else:                                                     # a batch: the first density and symmetric perturbations of it
        rngd = np.random.default_rng(1)
        dens = [P] + [P + 0.05 * (lambda B: B + B.T)(rngd.standard_normal((N, N))) for _ in range(nd - 1)]
    dP = torch.from_numpy(np.stack(dens)).to(dev)              # [nd, N, N], symmetric
    dJK = torch.zeros((2, nd, N, N), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def allreduce(t, op=None):
        kw = {} if op is None else {"op": op}
        if backend == "nccl":
            dist.all_reduce(t, **kw)
        else:                                                 # rehearsal: through the host
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)

    # The exchange step of a sharded build: inside the library (tf_comm_init: an RCCL communicator of the ranks, ncclAllReduce of J and K
    # issued by tf_fock_jk_device on the same stream, no host callback); TUNA_BENCH_TORCH_ALLREDUCE=1 or a gloo rehearsal: torch.distributed.
    in_library = world > 1 and backend == "nccl" and os.environ.get("TUNA_BENCH_TORCH_ALLREDUCE") is None
    if in_library:
        from tuna_amd import distributed as tdist
        try:
            tdist.attach_rccl(eng)
            ok_comm = 1.0
        except Exception as e:                                # (librccl not loadable on this box, ...): the torch all-reduce then -- on EVERY rank
            ok_comm = 0.0
            if rank == 0:
                print(f"bench.py: in-library RCCL communicator not available ({type(e).__name__}: {e}); using torch.distributed", file=sys.stderr)
        flag = torch.tensor([ok_comm], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if float(flag.item()) == 0.0:
            if eng.comm_attached():
                eng.comm_destroy()
            in_library = False

    def step():
        eng.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), nd, stream)
        if world > 1 and not in_library:
            allreduce(dJK)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.jk_profile(True)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_s, kern_n = eng.jk_profile_read()
    eng.jk_profile(False)
    t_max = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    k_max = torch.tensor([kern_s / max(kern_n, 1)], dtype=torch.float64, device=dev)
    if world > 1:
        allreduce(t_max, dist.ReduceOp.MAX)
        allreduce(k_max, dist.ReduceOp.MAX)
    elapsed = float(t_max.item())
    kernel_avg_s = float(k_max.item())

    # sanity of the timed result: J symmetric, tr-type identity <P|J> > 0
    Jh = dJK[0, 0].cpu().numpy()
    ok = bool(np.isfinite(Jh).all() and np.abs(Jh - Jh.T).max() < 1e-8 * max(1.0, np.abs(Jh).max()))

    if rank == 0:
        layout = st["layout"]
        alg_bytes = 8.0 * N ** 4 / world                    # 8 N^4 bytes per build, per GPU 8 N^4 / G
        stored_bytes = float(st["bytes"])
        traffic, traffic_src, traffic_meta = pmc_traffic(args.workload, world, layout, stored_bytes) if nd == 1 else (None, None, None)
        # roofline of the dominant kernel on the bytes it HAS to read: the stored tensor, once per build.  What the kernel moves on top
        # (density re-reads, partial sums) is waste and shows up as traffic_ratio > 1, not as achievement.
        achieved = stored_bytes / kernel_avg_s / 1e9
        storage = {"packed": "parity-blocked 8-fold symmetry-unique values of the spherical tensor: row (i>=j) keeps the pairs (k>=l) <= (i,j) "
                             "whose x/y reflection parity class equals that of (i,j) -- the others are exact zeros (pyx:1324-1327) -- f64, "
                             "units of 8 interleaved rows, sharded by (ij) shell pair over ranks",
                   "rows": "rows (i>=j) x full (k,l) of the spherical tensor, f64, sharded by (ij) shell pair over ranks",
                   "tiles": "the same 8-fold unique, parity-blocked values arranged for v_mfma_f64_16x16x4: per first index i and class pair, "
                            "(k,l) rectangles / triangles below i in strips of 64 x 16-column blocks, second index innermost (tf_tiles.h); "
                            "opt-in (DESIGN.md section 4.1b: slower than `packed` at N = 400)"}[layout]
        out = {
            "metric": "Fock builds/sec (J+K from the HBM-resident ERI tensor, one density) + SCF wall time",
            "value": args.steps / elapsed, "unit": "Fock builds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if backend == "nccl" or world == 1 else "synthetic (REHEARSAL: gloo, ranks sharing cards -- not a measurement)",
            "config": {"workload": f"{args.workload}: {desc}", "n_ao_spherical": N, "n_ao_cartesian": eng.n_cart,
                       "n_shells": eng.n_shell, "n_densities": nd, "layout": layout, "storage": storage,
                       "stored_bytes_per_gpu": stored_bytes, "parallelism": (f"ij-row shards x{world} + RCCL all-reduce of [J;K] " + ("issued by the library (tf_comm_init)" if in_library else "through torch.distributed")) if world > 1 else "1 GPU",
                       "result_ok": ok},
            # all densities of a step: Fock matrices per second, and the executed FP64 rate -- six multiply-adds per stored value and
            # density (tf_jkpacked.hip.h) -- against the FP64 peak (78.6 TFLOP/s; vector and matrix cores run FP64 at the same rate on CDNA4)
            "per_density": {"n_densities": nd, "fock_matrices_per_s": nd * args.steps / elapsed,
                            "executed_fp64_tflops": 12.0 * (stored_bytes / 8.0) * nd * args.steps / elapsed / 1e12,
                            "frac_of_fp64_peak": 12.0 * (stored_bytes / 8.0) * nd * args.steps / elapsed / FP64_VECTOR_PEAK_FLOPS,
                            "passes_over_the_tensor_per_step": (nd + 7) // 8 if layout == "tiles" else (nd + 1) // 2,
                            "mfma": pmc_mfma(args.workload, nd, kernel_avg_s) if layout == "tiles" and nd >= 2 else None},
            # roofline of the dominant kernel on PHYSICAL bytes (a fraction of the 8 TB/s HBM peak, <= 1); the reference's dense 8 N^4
            # bytes per build that the same launch stands for are reported separately
            "roofline": {"bound": "hbm", "kernel": JK_KERNEL[layout], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_ratio": (traffic / stored_bytes) if traffic else None, "traffic_source": traffic_src,
                         "traffic_profile": traffic_meta,
                         "kernel_avg_ms": 1e3 * kernel_avg_s, "necessary_bytes_per_launch": stored_bytes,
                         "frac_on_ms_per_step": stored_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_equivalent": {"bytes_per_launch": alg_bytes, "GBs": alg_bytes / kernel_avg_s / 1e9,
                                                    "reduction_vs_dense": alg_bytes / stored_bytes,
                                                    "permutational_symmetry": 8.0 if layout == "packed" else 2.0,
                                                    "parity_zeros_and_padding": alg_bytes / stored_bytes / (8.0 if layout == "packed" else 2.0),
                                                    "note": "SURVEY 8d prices a build at the reference's dense 8 N^4 bytes; the kernel streams "
                                                            "1/reduction_vs_dense of them (8-fold permutational symmetry x the x/y parity rule, less padding)"},
                         "note": "achieved = stored tensor bytes (each read once per build: the bytes the build cannot avoid) / average kernel time "
                                 "from HIP events on the launch stream; traffic = HBM bytes per build from the PMC counters of the committed profile "
                                 "(FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md section HBM); traffic_ratio = traffic / stored bytes"},
            "eri_build": {"wall_s": eri_wall, "cold_wall_s": eri_wall_cold, "device_s": {k: float(v) for k, v in eri_t.items() if k.endswith("_s")},
                          "shell_quartets": eri_t["shell_quartets"], "primitive_shell_quartets": eri_t["primitive_shell_quartets"],
                          "component_quartets": eri_t["component_quartets"],
                          "component_quartets_per_s": eri_t["component_quartets"] / max(eri_t["total_s"], 1e-12),
                          "nominal_flops": eri_t.get("nominal_flops"),
                          "gflops": (eri_t["nominal_flops"] / max(eri_t["cart_kernel_s"], 1e-12) / 1e9) if eri_t.get("nominal_flops") else None,
                          "frac_of_fp64_vector_peak": (eri_t["nominal_flops"] / max(eri_t["cart_kernel_s"], 1e-12) / FP64_VECTOR_PEAK_FLOPS)
                          if eri_t.get("nominal_flops") else None,
                          "executed": pmc_eri_valu(args.workload, eri_t["cart_kernel_s"]),
                          "flops_note": "nominal_flops = the reference algorithm's count for the quartets evaluated (SURVEY 8d(ii): per primitive AO "
                                        "quartet 8 x inner terms of the loop nest pyx:1179-1217 + 6 (L+1) + 3 (L+1)^2 / 2 + 60), divided by the time of "
                                        "the ERI kernels; the kernels factorise the sum per shell quartet and execute fewer operations than that",
                          "note": "quartets actually evaluated on this rank (packed layout: ket shell pairs up to the bra's first shell); "
                                  "FP64-vector / latency-bound work (DESIGN.md section 4.2), not priced against HBM or MFMA"},
        }
    # SCF wall time, the second half of the metric: collective when the tensor is sharded (every rank runs the native cycle on its
    # rows; one all-reduce of the partial [J;K] per Fock build through torch.distributed / RCCL), so every rank takes part
    if not args.no_scf:
        if world > 1 and not eng.comm_attached():
            from tuna_amd import distributed as tdist
            tdist.attach_allreduce(eng)
        # (an exception in a leg must not cost the headline line: it is reported in the leg's place.  With several ranks a leg that
        # fails on one rank fails on all of them -- the flag is all-reduced -- so that no rank is left waiting in a collective)
        def run_leg(fn):
            try:
                res = fn()
                ok_leg = 1.0 if "error" not in res else 0.0
            except Exception as e:
                res, ok_leg = {"error": f"{type(e).__name__}: {e}"}, 0.0
            if world > 1:
                flag = torch.tensor([ok_leg], dtype=torch.float64, device=dev)
                allreduce(flag, dist.ReduceOp.MIN)
                if float(flag.item()) == 0.0 and "error" not in res:
                    res = {"error": "the leg failed on another rank"}
            return res
        scf_w = run_leg(lambda: scf_on_workload(eng, atoms, shells, nocc, desc))   # the tensor of the timed builds is still resident
        scf_c = run_leg(lambda: scf_leg(eng, args, rank, world, allreduce))
        if rank == 0:
            out["scf_on_workload"], out["scf"] = scf_w, scf_c
    # the CPU baseline comes LAST: its 16 BLAS threads keep spinning for a while after the einsums and, inside the box's CPU quota, throttle
    # the HIP runtime threads of whatever GPU leg follows (the allocation-heavy MP2 leg measured 0.5 s instead of 0.055 s)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_fock(N)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def scf_on_workload(eng, atoms, shells, nocc, desc):
    """RHF wall time on the bench workload itself (tensor already resident): what one SCF iteration costs at the north-star size --
    Fock build, DIIS algebra, eigenvector refinement (tf_scf.hip.h).  Core guess, TIGHT thresholds, DIIS 6, no damping."""
    from tuna_amd import molecule as mol
    from tuna_amd._lib import TunaError
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    res = {"workload": desc}
    try:
        best = None
        for _ in range(2):      # the first pass pays the one-time rocSOLVER start-up
            t0 = time.perf_counter()
            S, T, V, _, _ = eng.one_electron(xyz, chg, [0, 0, 0.5 * atoms[-1].origin[2]])
            X, smin, _ = eng.orthogonaliser(S)
            _, C0 = eng.diagonalise(T + V, X)
            P0 = 2.0 * C0[:, :nocc] @ C0[:, :nocc].T
            P0 = 0.5 * (P0 + P0.T)
            t1 = time.perf_counter()
            nao = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
            # The core guess of this Ar2-like system cuts through its near-degenerate n = 3 shell (2.4e-6 Eh between the last occupied
            # and the first empty orbital), so the undamped cycle is sensitive to the last digits of the guess: the same code has taken
            # 18 or 24 iterations or none at all depending on the eigensolver's rounding.  Undamped first (the configuration of rounds
            # 1-3); if that does not converge, the reference's default dynamic damping -- and the line says which one ran.
            damping = "none"
            try:
                r = eng.scf_rhf(S, T, V, P0, float(np.sum(P0 * (T + V))), nocc, mol.nuclear_repulsion(atoms), X=X, conv="tight", damping=damping,
                                n_atom_ao=nao, max_iter=100)
            except TunaError:
                damping = "dynamic (the undamped cycle did not converge from this core guess)"
                t1 = time.perf_counter()
                r = eng.scf_rhf(S, T, V, P0, float(np.sum(P0 * (T + V))), nocc, mol.nuclear_repulsion(atoms), X=X, conv="tight", damping="dynamic",
                                n_atom_ao=nao, max_iter=200)
            t2 = time.perf_counter()
            orbitals = (r["C"], r["epsilons"])
            best = {"energy_Eh": r["energy"], "iterations": r["n_iter"], "damping": damping, "eigensolver_paths": eng.eigh_stats(), "fock_paths": eng.jk_path_stats(),
                    "setup_wall_s": t1 - t0, "scf_wall_s": t2 - t1,
                    "ms_per_iteration": 1e3 * (t2 - t1) / r["n_iter"], "fock_kernels_ms_per_iteration": 1e3 * r["fock_seconds"] / r["n_iter"],
                    "eigen_ms_per_iteration": 1e3 * r["eig_seconds"] / r["n_iter"], "smallest_overlap_eigenvalue": smin}
        res.update(best)
        if eng.world == 1:
            res["mp2"] = mp2_leg(eng, orbitals[0], orbitals[1], nocc)
    except TunaError as e:      # e.g. a synthetic basis too linearly dependent for an SCF: the Fock-build numbers stand on their own
        res["error"] = str(e)
    return res


def mp2_leg(eng, C, eps, nocc):
    """The GEMM-shaped consumer of the resident tensor (BASELINE config 5's step at the workload size):
    AO->MO transformation of the (ia|jb) block + RMP2 energy with the converged orbitals, priced against the FP64 MATRIX peak."""
    N = eng.N
    o, v = nocc, N - nocc
    for _ in range(3):                                            # (rocBLAS kernel selection, the work-space pool; measured: the first three calls of a process are slow)
        eng.mp2_rhf(C, eps, nocc)
    r = min((eng.mp2_rhf(C, eps, nocc) for _ in range(3)), key=lambda x: x["seconds"])
    rows = N * (N + 1) // 2
    # executed multiply-adds x 2: the first quarter works class by class on the four nonzero blocks of a row ((k of class a) x (l of
    # class a ^ c): sum_a |a| |a ^ c| products per occupied orbital instead of N^2); the x/y parity class of an output AO is that of the
    # first Cartesian component of its row of the spherical matrix
    U, lmn = eng.sph_matrix(), np.asarray(eng_aos_lmn(eng))
    first = np.argmax(np.abs(U) > 0, axis=1)
    cls = (lmn[first, 0] & 1) | ((lmn[first, 1] & 1) << 1)
    size = np.bincount(cls, minlength=4).astype(float)
    hi, lo = np.tril_indices(N)
    rows_c = np.bincount(cls[hi] ^ cls[lo], minlength=4).astype(float)
    q1_blocks = sum(rows_c[c] * 2.0 * o * sum(size[a] * size[a ^ c] for a in range(4)) for c in range(4))
    flops_r2 = q1_blocks + rows * 2.0 * o * N * v + 2.0 * o * N * N * o * v + o * 2.0 * v * N * o * v      # round 2's order (expanded blocks)
    dense = rows * (2.0 * o * N * N + 2.0 * o * N * v) + 2.0 * o * N * N * o * v + o * 2.0 * v * N * o * v
    # round 3 (tf_mp2.hip.h: mo_q1_kernel + transform_q1): the SHORT index first.  Executed multiply-adds x 2: first quarter on the packed
    # segments -- every stored value feeds both of its images, o columns each --, then mu -> i (K = N), nu -> a, sigma -> b
    stored = eng.eri_storage()["bytes"] / 8.0
    flops = 4.0 * o * stored + 2.0 * o * N * (N * N * o) + o * 2.0 * v * N * (N * o) + o * v * 2.0 * v * o * N
    fast = o <= 32 and os.environ.get("TF_MO_Q1", "1") != "0" and eng.eri_storage()["layout"] == "packed"
    if not fast:
        flops = flops_r2
    return {"E_MP2_Eh": r["E_MP2"], "seconds": r["seconds"], "flops": flops, "tflops": flops / r["seconds"] / 1e12,
            "frac_of_fp64_matrix_peak": flops / r["seconds"] / FP64_MATRIX_PEAK_FLOPS,
            "algorithm": "short index first: hand-written MFMA-f64 first quarter on the packed segments, then three rocBLAS GEMMs" if fast
                         else "round-2 order: expanded parity blocks, rocBLAS only",
            "flops_round2_order": flops_r2, "tflops_round2_order_equivalent": flops_r2 / r["seconds"] / 1e12,
            "flops_if_rows_were_dense": dense, "tflops_dense_equivalent": dense / r["seconds"] / 1e12,
            "note": "ovov-only transformation on the stored (i >= j) rows; flops = EXECUTED operations of the path that ran: 4 o x stored values "
                    "(both images of every stored pair (kl) <= (ij), o occupied columns: mo_q1_kernel on v_mfma_f64_16x16x4_f64, C as the B operand) "
                    "+ 2 o^2 N^3 + 2 o^2 v N^2 + 2 o^2 v^2 N (rocBLAS); flops_round2_order = what round 2's expanded-block order executed for the "
                    "same result (sum_c rows_c 2 o sum_a |a||a^c| + rows 2 o N v + ...)"}


def eng_aos_lmn(eng):
    """Cartesian exponents (lx, ly, lz) of the engine's current basis, from the AO list it was given."""
    return eng.aos.lmn


def scf_leg(eng, args, rank=0, world=1, allreduce=None):
    """SCF wall time on BASELINE.json configs[1] (N2 RHF/cc-pVTZ): ERI build + native RHF (EXTREME thresholds, core guess,
    DIIS 6, no damping) on the GPU, energy checked against the reference anchor; CPU ERI build beside it."""
    from tuna_amd import molecule as mol
    atoms, shells, aos, nocc, desc = build_workload("n2-cc-pvtz")
    xyz, chg = [a.origin for a in atoms], [float(a.charge) for a in atoms]
    passes = []
    for _ in range(3):          # pass 0 is cold (rocBLAS/rocSOLVER start-up, first allocations); the last one is reported as warm
        t0 = time.perf_counter()
        eng.set_basis(aos).build_eri(True)
        t_eri = time.perf_counter() - t0
        t1 = time.perf_counter()
        S, T, V, _, _ = eng.one_electron(xyz, chg, [0, 0, 0.5 * atoms[1].origin[2]])
        X, _, _ = eng.orthogonaliser(S)
        _, C0 = eng.diagonalise(T + V, X)
        P0 = 2.0 * C0[:, :nocc] @ C0[:, :nocc].T
        P0 = 0.5 * (P0 + P0.T)
        E0 = float(np.sum(P0 * (T + V)))
        r = eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="none", n_atom_ao=[30, 30])
        t_scf = time.perf_counter() - t1
        passes.append((t_eri, t_scf))
    cold_eri, cold_scf = passes[0]
    # Fock builds/s on this config too (device-resident P)
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    dP = torch.from_numpy(r["P"]).to(dev)
    dJK = torch.zeros((2, eng.N, eng.N), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one_build():
        eng.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), 1, stream)
        if world > 1 and not eng.comm_attached():
            allreduce(dJK)
    for _ in range(5):
        one_build()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(200):
        one_build()
    torch.cuda.synchronize()
    fps = 200 / (time.perf_counter() - t2)
    out = {"workload": desc, "energy_Eh": r["energy"], "abs_error_vs_reference_anchor_Eh": abs(r["energy"] - ANCHOR_N2_CCPVTZ),
           "iterations": r["n_iter"], "eri_build_wall_s": t_eri, "scf_wall_s": t_scf, "total_wall_s": t_eri + t_scf,
           "cold_first_call": {"eri_build_wall_s": cold_eri, "scf_wall_s": cold_scf, "total_wall_s": cold_eri + cold_scf},
           "note": "wall times of the 3rd repetition in this process (1e integrals + orthogonaliser + core guess + EXTREME RHF in scf_wall_s)",
           "fock_kernel_s": r["fock_seconds"], "eigensolver_s": r["eig_seconds"], "fock_builds_per_s": fps}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_eri_build"] = cpu_baseline_eri(aos)
        out.update(cpu_baseline_scf(aos, shells, atoms, nocc, S, T, V, X, P0, E0, r["energy"], r["n_iter"]))
    return out


def cpu_baseline_scf(aos, shells, atoms, nocc, S, T, V, X, P0, E0, gpu_energy, gpu_iters):
    """The CPU path timed on this box for the same input (N2 RHF/cc-pVTZ, EXTREME, core guess, DIIS 6, no damping): the C port's ERI build
    (OpenMP over AO pairs, all CPUs of the process), the dense Cartesian -> spherical transform and the NumPy restatement of the
    reference's RHF cycle with its einsum Fock builds (oracle/scf_oracle.py; scf:1072-1154, 1292-1435)."""
    from oracle import oracle as orc
    from oracle import scf_oracle as so
    from tuna_amd import molecule as mol
    from tuna_amd import spherical
    import tuna_amd
    cores = tuna_amd.cpu_quota()
    t0 = time.perf_counter()
    E_cart = orc.eri(aos, cores)
    t_eri = time.perf_counter() - t0
    U = spherical.transformation_matrix([sh.L for sh in shells])
    t0 = time.perf_counter()
    E_sph = so.eri_to_spherical(U, E_cart)
    t_sph = time.perf_counter() - t0
    t0 = time.perf_counter()
    o = so.run_rhf(S, T, V, E_sph, X, P0, E0, nocc, mol.nuclear_repulsion(atoms), [30, 30], conv="extreme", damping=False)
    t_scf = time.perf_counter() - t0
    return {"cpu_scf_wall_s": t_scf, "cpu_scf": {"kind": "port", "cores": int(cores), "eri_build_s": t_eri, "spherical_transform_s": t_sph,
                                                 "scf_loop_s": t_scf, "total_s": t_eri + t_sph + t_scf, "iterations": o["n_iter"],
                                                 "fock_builds_per_s": o["n_iter"] / t_scf if t_scf > 0 else None,
                                                 "energy_Eh": o["energy"], "abs_diff_to_gpu_energy_Eh": abs(o["energy"] - gpu_energy),
                                                 "gpu_iterations": gpu_iters}}


if __name__ == "__main__":
    main()
