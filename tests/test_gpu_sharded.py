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
            J, K = eng.fock_jk(g["P_rand"])                            # partial sums over this rank's rows
            jk = torch.from_numpy(np.stack([J, K]))
            tdist.all_reduce_jk_(jk)
            Jf, Kf = jk.numpy()
            # a sample of the tensor: rows owned elsewhere read as zero here, the sum over ranks is the reference value
            v = torch.from_numpy(eng.sample_eri(g["eri_sph_idx"][:2000]))
            dist.all_reduce(v)
            ret[rank] = (float(np.abs(Jf - g["J_rand"]).max()), float(np.abs(Kf - g["K_rand"]).max()),
                         float(np.abs(v.numpy() - g["eri_sph_val"][:2000]).max()), st["bytes"])
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
