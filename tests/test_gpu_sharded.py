"""GPU: the sharded (world > 1) path of the library on ONE card -- two processes, each a rank with its own context and its
own half of the tensor rows on cuda:0; partial J/K are summed with a gloo all-reduce on host copies (RCCL refuses two ranks on
one device; the bench uses RCCL on real multi-GPU nodes).  Also the row-length variants of the J/K kernel that single-GPU
sizes never reach."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, tag, mode, layout, ret):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    if mode:
        os.environ["TF_ERI_MODE"] = mode
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import make_system
        from tuna_amd.engine import Engine
        from tuna_amd import distributed as tdist
        atoms, shells, aos, nocc = make_system(tag)
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", tag + ".npz"))
        with Engine(0, rank, world) as eng:
            eng.set_basis(aos).build_eri(True, layout=layout)
            st = eng.eri_storage()
            rows = st["rows"]
            assert st["layout"] == layout
            owner = tdist.row_owner_matrix(shells, world, layout=layout)
            assert rows == int((owner == rank).sum())                 # the library followed the shared plan
            fock = tdist.ShardedFock(eng)                              # partial sums over this rank's rows + ONE all-reduce
            Jf, Kf = fock(g["P_rand"])
            # two densities in one fused pass (what a UHF build issues): the second one a scaled, shifted copy
            P2 = np.stack([g["P_rand"], 0.5 * g["P_rand"] + 0.25 * np.diag(np.diag(g["P_rand"]))])
            J2, K2 = fock(P2)
            assert np.abs(J2[0] - Jf).max() < 1e-10 and np.abs(K2[0] - Kf).max() < 1e-10
            Jd, Kd = fock(P2[1])
            assert np.abs(J2[1] - Jd).max() < 1e-10 and np.abs(K2[1] - Kd).max() < 1e-10
            # a sample of the tensor: rows owned elsewhere read as zero here, the sum over ranks is the reference value
            v = torch.from_numpy(eng.sample_eri(g["eri_sph_idx"][:2000]))
            dist.all_reduce(v)
            ret[rank] = (float(np.abs(Jf - g["J_rand"]).max()), float(np.abs(Kf - g["K_rand"]).max()),
                         float(np.abs(v.numpy() - g["eri_sph_val"][:2000]).max()), st["bytes"])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,tag,mode,layout", [(2, "c3_ar2_ccpvqz", "", "packed"), (4, "c3_ar2_ccpvqz", "", "packed"), (4, "n2_ccpvdz", "class", "packed")])
def test_config3_ar2_ccpvqz_sharded_and_four_ranks(world, tag, mode, layout):
    """BASELINE config 3 verbatim (Ar2 RHF/cc-pVQZ, shell-pair shards) on two and on four ranks sharing the card; four ranks also with the
    per-class ERI kernels on a small system (ranks whose row share of a class is empty)."""
    import torch.multiprocessing as mp
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rank_main, args=(world, _free_port(), tag, mode, layout, ret), nprocs=world, join=True)
        res = dict(ret)
    assert set(res) == set(range(world))
    for eJ, eK, eV, nbytes in res.values():
        assert eJ < 1e-9 and eK < 1e-9 and eV < 1e-12
    n_total = sum(v[3] for v in res.values())
    assert max(v[3] for v in res.values()) - min(v[3] for v in res.values()) <= 0.08 * n_total


def test_whole_shell_plan_numerics(monkeypatch):
    """The whole-shell plan (TF_SHARD_PLAN=shells: what a large problem gets by default) forced on BASELINE config 3 with two ranks: a rank
    holds complete runs of j for its rows -- J, K and tensor samples against the reference's golden values on both ranks (the balance of
    34 shells over two ranks is not asserted: the default would fall back to the segment plan if it were off by more than 3 %)."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("TF_SHARD_PLAN", "shells")
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rank_main_plain, args=(2, _free_port(), "c3_ar2_ccpvqz", ret), nprocs=2, join=True)
        res = dict(ret)
    assert set(res) == {0, 1}
    for eJ, eK, eV in res.values():
        assert eJ < 1e-9 and eK < 1e-9 and eV < 1e-12


def _rank_main_plain(rank, world, port, tag, ret):
    """A rank of a sharded build under whatever plan the environment selects: Fock build + tensor samples against the golden values."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import make_system
        from tuna_amd.engine import Engine
        from tuna_amd import distributed as tdist
        atoms, shells, aos, nocc = make_system(tag)
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", tag + ".npz"))
        with Engine(0, rank, world) as eng:
            eng.set_basis(aos).build_eri(True)
            owner = tdist.row_owner_matrix(shells, world)
            assert eng.eri_storage()["rows"] == int((owner == rank).sum())
            Jf, Kf = tdist.ShardedFock(eng)(g["P_rand"])
            v = torch.from_numpy(eng.sample_eri(g["eri_sph_idx"][:2000]))
            dist.all_reduce(v)
            ret[rank] = (float(np.abs(Jf - g["J_rand"]).max()), float(np.abs(Kf - g["K_rand"]).max()),
                         float(np.abs(v.numpy() - g["eri_sph_val"][:2000]).max()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["packed", "rows"])
@pytest.mark.parametrize("tag,mode", [("n2_ccpvdz", ""), ("c2_n2_ccpvtz", "class"), ("c4_co_def2tzvp", "generic")])
def test_two_ranks_on_one_card(tag, mode, layout):
    import torch.multiprocessing as mp
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rank_main, args=(world, _free_port(), tag, mode, layout, ret), nprocs=world, join=True)
        res = dict(ret)
    assert set(res) == {0, 1}
    for eJ, eK, eV, nbytes in res.values():
        assert eJ < 1e-10 and eK < 1e-10 and eV < 1e-12
    n_total = res[0][3] + res[1][3]
    assert abs(res[0][3] - res[1][3]) <= 0.05 * n_total                # the plan balances the stored bytes


def _rank_scf(rank, world, port, kind, tag, ret):
    """One rank of a native SCF cycle on a sharded tensor: tf_scf_rhf / tf_scf_uhf with the registered all-reduce of the partial
    [J;K] (tuna_amd.distributed.attach_allreduce; gloo here, two ranks sharing one card)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import UHF_SYSTEMS, atom_arrays, make_system
        from oracle import scf_oracle as so
        from tuna_amd import distributed as tdist, molecule as mol
        from tuna_amd.engine import Engine, SCF_CONVERGENCE
        with Engine(0, rank, world) as eng:
            if kind == "rhf":
                atoms, shells, aos, nocc = make_system(tag)
                eng.set_basis(aos).build_eri(True)
                xyz, chg, org = atom_arrays(atoms)
                S, T, V, _, _ = eng.one_electron(xyz, chg, org, spherical=True)
                X, _, _ = eng.orthogonaliser(S)
                P0, E0 = so.core_guess(T, V, X, nocc)
                ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
                try:
                    eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="dynamic", n_atom_ao=ranges)
                    refused = False
                except Exception as e:                               # no all-reduce registered yet: partial sums must not be used
                    refused = "tf_set_allreduce" in str(e)
                tdist.attach_allreduce(eng)
                r = eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="extreme", damping="dynamic", n_atom_ao=ranges)
                ret[rank] = (refused, r["energy"], r["n_iter"], r["table"], r["epsilons"])
            else:
                from tuna_amd import scf
                from tuna_amd.energy import Calculation, build_molecule_and_integrals
                sym, R, basis, na, nb = UHF_SYSTEMS[tag]
                calc = Calculation(basis=basis, SCF_conv=SCF_CONVERGENCE["extreme"], multiplicity=na - nb + 1, damping=False, core_guess=True)
                molecule, integrals, X, guess, _ = build_molecule_and_integrals(sym, R, calc, eng)
                out = scf.run_self_consistent_field_cycle(molecule, calc, integrals, 0.0, X, guess)   # attaches the all-reduce itself
                ret[rank] = (True, out.energy, out.n_iterations, out.table, np.concatenate((out.epsilons_alpha, out.epsilons_beta)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,tag,plan", [("rhf", "n2_ccpvdz", ""), ("rhf", "c4_co_def2tzvp", ""), ("uhf", "oh_doublet_ccpvdz", ""),
                                           ("rhf", "c4_co_def2tzvp", "shells")])
def test_native_scf_cycle_on_a_sharded_tensor(kind, tag, plan, golden, uhf_golden, monkeypatch):
    """The native cycles with the tensor split over two ranks: per iteration ONE all-reduce of the stacked [J;K], everything else on
    the device of each rank -- same trajectory as the reference run (golden) on every rank."""
    import torch.multiprocessing as mp
    world = 2
    if plan:
        monkeypatch.setenv("TF_SHARD_PLAN", plan)                      # (the whole-shell plan of large problems, forced on a small one)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rank_scf, args=(world, _free_port(), kind, tag, ret), nprocs=world, join=True)
        res = dict(ret)
    assert set(res) == {0, 1}
    if kind == "rhf":
        g = golden(tag)
        ref_E, ref_table = float(g["scf_energy"]), g["scf_table"]
    else:
        g = uhf_golden[tag]
        ref_E, ref_table = float(g["scf_energy_nodamp"]) - float(g["V_NN"]), g["scf_table_nodamp"]
    for rank, (refused, E, n_iter, table, eps) in res.items():
        assert refused                                               # without the hook a sharded tensor is refused
        assert abs(E - ref_E) < 1e-8
        assert abs(n_iter - len(ref_table)) <= 1
    assert abs(res[0][1] - res[1][1]) < 1e-10 and res[0][2] == res[1][2]
    np.testing.assert_allclose(res[0][3][:, 1], res[1][3][:, 1], atol=1e-9)          # both ranks walk the same trajectory
    n = min(res[0][2], len(ref_table))
    if kind == "rhf":
        np.testing.assert_allclose(res[0][3][:n, 1], ref_table[:n, 1], atol=1e-8)
        np.testing.assert_allclose(res[0][3][:n, 6], ref_table[:n, 6], atol=1e-6)    # damping factors


def _nccl_single_rank(rank, port, ret):
    """RCCL itself: a ONE-rank `nccl` process group (RCCL refuses two ranks on one device, so one rank is what a one-GPU box can run) around
    rank 0's context of a TWO-way shard.  The registered hook then really issues ncclAllReduce on the library's buffer and stream --
    the sum over a one-rank group is the rank's own partial [J;K], which is what is checked."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from conftest import atom_arrays, make_system
        from oracle import scf_oracle as so
        from tuna_amd import distributed as tdist, molecule as mol
        from tuna_amd._lib import TunaError
        from tuna_amd.engine import Engine
        atoms, shells, aos, nocc = make_system("n2_ccpvdz")
        g = np.load(os.path.join(os.path.dirname(__file__), "golden", "n2_ccpvdz.npz"))
        with Engine(0, 0, 2) as eng:                                   # rank 0 of 2: half of the rows
            eng.set_basis(aos).build_eri(True)
            N = eng.N
            dev = torch.device("cuda", 0)
            dP = torch.from_numpy(np.ascontiguousarray(g["P_rand"])).to(dev)
            dJK = torch.zeros((2, N, N), dtype=torch.float64, device=dev)
            eng.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
            partial = dJK.cpu().numpy().copy()
            dist.all_reduce(dJK)                                        # what bench.py --gpus N issues per build
            after = dJK.cpu().numpy()
            # the exchange step INSIDE the library (tf_comm_init: librccl loaded by the library, ncclAllReduce issued by allreduce_jk on the
            # build's stream -- no Python frame per build): first on the Fock build itself ...
            tdist.attach_rccl(eng)
            assert eng.comm_attached()
            eng.fock_jk_device(dP.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            inlib = dJK.cpu().numpy().copy()
            assert np.array_equal(inlib, partial), "the sum over a one-rank communicator is the rank's own partial sums"
            # ... then in the native cycle: every Fock build of it all-reduces [J;K] and the agreement vector through the communicator
            xyz, chg, org = atom_arrays(atoms)
            S, T, V, _, _ = eng.one_electron(xyz, chg, org, spherical=True)
            X, _, _ = eng.orthogonaliser(S)
            P0, E0 = so.core_guess(T, V, X, nocc)
            ranges = [sum(s.n_sph for s in shells if s.atom == a) for a in range(len(atoms))]
            try:
                r = eng.scf_rhf(S, T, V, P0, E0, nocc, mol.nuclear_repulsion(atoms), X=X, conv="loose", damping="none", n_atom_ao=ranges, max_iter=4)
                outcome = ("ran", r["n_iter"])
            except TunaError as e:                                     # half a tensor does not have to converge in four iterations
                outcome = ("error", str(e))
            # the same first Fock build through the hook-less path: J, K of P0 from this rank's rows
            dP0 = torch.from_numpy(np.ascontiguousarray(P0)).to(dev)
            eng.fock_jk_device(dP0.data_ptr(), dJK[0].data_ptr(), dJK[1].data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
            ret[0] = (float(np.abs(partial - after).max()), float(np.abs(partial).max()), outcome, dist.get_backend(), bool(np.isfinite(dJK.cpu().numpy()).all()))
    finally:
        dist.destroy_process_group()


def test_rccl_branch_of_the_exchange_step_runs_on_a_single_rank_group():
    """The RCCL branches -- bench.py's torch all-reduce of [J;K], and the library's own communicator (tf_comm_init: the exchange step of
    tf_fock_jk_device and of the native cycles without a host callback) -- executed on hardware."""
    import torch.multiprocessing as mp
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_nccl_single_rank, args=(_free_port(), ret), nprocs=1, join=True)
        res = dict(ret)
    diff, scale, outcome, backend, finite = res[0]
    assert backend == "nccl" and finite
    assert diff == 0.0 and scale > 0.0                               # the sum over a one-rank communicator is the rank's own partial sums
    # the native cycle went through the communicator (ncclAllReduce of [J;K], then of the agreement vector); its agreement test then
    # notices, correctly, that the one-rank group sums ONE rank's decision values where the context expects the sum over two
    assert outcome[0] == "error" and "disagree on a control decision" in outcome[1], outcome


def _rank_mp2(rank, world, port, tag, layout, ret):
    """One rank of an RMP2 / AO->MO transformation on a sharded tensor: every rank transforms the rows it owns, one all-reduce of the
    transformed tensor (the hook of attach_allreduce)."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import make_system
        from tuna_amd import distributed as tdist
        from tuna_amd.engine import Engine
        gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "mp2_systems.npz"))
        g = {k.split("__", 1)[1]: gold[k] for k in gold.files if k.startswith(tag + "__")}
        atoms, shells, aos, nocc = make_system({"n2_sto3g": "n2_sto3g", "n2_ccpvdz": "n2_ccpvdz", "c5_n2_ccpvtz": "c2_n2_ccpvtz"}[tag])
        with Engine(0, rank, world) as eng:
            eng.set_basis(aos).build_eri(True, layout=layout)
            try:
                eng.mp2_rhf(g["C"], g["eps"], nocc)
                refused = False
            except Exception as e:                                   # partial sums must not be returned as the result
                refused = "tf_set_allreduce" in str(e)
            tdist.attach_allreduce(eng)
            r = eng.mp2_rhf(g["C"], g["eps"], nocc)
            rng = np.random.default_rng(3)
            N = eng.N
            Cs = [rng.standard_normal((N, n)) for n in (2, 3, 4, 2)]
            mixed = eng.ao_to_mo(*Cs)
            same = eng.ao_to_mo(g["C"]) if N <= 30 else None
            ret[rank] = (refused, r["E_OS"], r["E_SS"], mixed, same)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tag,layout", [("n2_sto3g", "packed"), ("n2_ccpvdz", "packed"), ("n2_ccpvdz", "rows"), ("c5_n2_ccpvtz", "packed")])
def test_mp2_on_a_sharded_tensor(tag, layout, engine, mp2_golden):
    """RMP2 (BASELINE config 5) and a mixed-block AO->MO transformation with the tensor split over two ranks equal the reference's
    values and the one-GPU transformation."""
    import torch.multiprocessing as mp
    from conftest import make_system
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rank_mp2, args=(world, _free_port(), tag, layout, ret), nprocs=world, join=True)
        res = dict(ret)
    assert set(res) == {0, 1}
    g = mp2_golden[tag]
    atoms, shells, aos, nocc = make_system({"n2_sto3g": "n2_sto3g", "n2_ccpvdz": "n2_ccpvdz", "c5_n2_ccpvtz": "c2_n2_ccpvtz"}[tag])
    engine.set_basis(aos).build_eri(True)
    rng = np.random.default_rng(3)
    Cs = [rng.standard_normal((engine.N, n)) for n in (2, 3, 4, 2)]
    one_gpu = engine.ao_to_mo(*Cs)
    for rank, (refused, e_os, e_ss, mixed, same) in res.items():
        assert refused
        assert abs(e_os - float(g["E_OS"])) < 1e-10 and abs(e_ss - float(g["E_SS"])) < 1e-10
        assert np.abs(mixed - one_gpu).max() < 1e-11 * max(1.0, np.abs(one_gpu).max())
        if same is not None:
            idx = g["mo_idx"]
            assert np.abs(same[idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]] - g["mo_val"]).max() < 1e-11


def test_jk_kernel_variants_agree(golden):
    """jk_rows_kernel<NLC,JB,ND> instantiations that production sizes on one GPU never select (NLC = 2, 4: N > 512) must give
    the same J/K bit for bit as the default one on a small tensor (TF_JK_FORCE_NLC / TF_JK_FORCE_JB test hooks)."""
    import subprocess, sys, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import make_system
from tuna_amd.engine import Engine
atoms, shells, aos, nocc = make_system("n2_ccpvdz")
g = np.load(%r)
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True, layout="rows")
    J, K = eng.fock_jk(np.stack([g["P_rand"], g["P_rand"].T * 0.5 + 0.1]))
print(json.dumps([float(np.abs(J[0] - g["J_rand"]).max()), float(np.abs(K[0] - g["K_rand"]).max()), float(J.sum()), float(K.sum())]))
''' % (os.path.join(os.path.dirname(__file__), ".."), os.path.dirname(__file__), os.path.join(os.path.dirname(__file__), "golden", "n2_ccpvdz.npz"))
    results = {}
    for nlc in ("", "2", "4"):
        for jb in ("", "1", "2", "4"):
            env = dict(os.environ)
            if nlc: env["TF_JK_FORCE_NLC"] = nlc
            if jb: env["TF_JK_FORCE_JB"] = jb
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            assert out.returncode == 0, out.stderr[-2000:]
            results[(nlc, jb)] = json.loads(out.stdout.strip().splitlines()[-1])
    ref = results[("", "")]
    assert ref[0] < 1e-10 and ref[1] < 1e-10
    for key, r in results.items():
        assert r[0] < 1e-10 and r[1] < 1e-10, key
        assert r[2] == ref[2], key                                 # J is summed in the same fixed tree for every variant
        assert abs(r[3] - ref[3]) < 1e-9 * abs(ref[3]), key
