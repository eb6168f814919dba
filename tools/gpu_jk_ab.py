"""GPU: A/B timing of the Fock build between library variants (tools/build_variant.sh), each in its own process, alternating.
usage: python tools/gpu_jk_ab.py [N] name1 name2 ...   ('base' = tuna_amd/libtunafock.so)"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
from tuna_amd import molecule as mol
from tuna_amd.engine import Engine
n = int(sys.argv[2])
counts = mol.synthetic_counts(n)
atoms = mol.make_atoms(["AR", "AR"], 7.1)
aos = mol.expand_cartesian_aos(mol.build_shells(atoms, {18: mol.even_tempered_basis(*counts)}))
rng = np.random.default_rng(0)
with Engine(0) as eng:
    eng.set_basis(aos).build_eri(True)
    t0 = time.perf_counter(); eng.build_eri(True); eri = time.perf_counter() - t0
    A = rng.standard_normal((eng.N, eng.N)); P = A + A.T
    for _ in range(5): eng.fock_jk(P)
    eng.jk_profile(True)
    t0 = time.perf_counter()
    for _ in range(30): J, K = eng.fock_jk(P)
    wall = (time.perf_counter() - t0) / 30
    ks, kn = eng.jk_profile_read()
    print(json.dumps({"kernel_ms": 1e3 * ks / kn, "build_ms_host_buffers": 1e3 * wall, "eri_s": eri, "GB": eng.eri_storage()["bytes"] / 1e9,
                      "checksum": float(np.sum(J * P) + np.sum(K * P))}))
"""
args = sys.argv[1:]
n = 400
if args and args[0].isdigit():
    n = int(args.pop(0))
names = args or ["base"]
for rep in range(2):
    for name in names:
        env = dict(os.environ)                               # name = a library variant, or VAR=value for the base library with that variable
        if "=" in name:
            k, v = name.split("=", 1)
            env[k] = v
        lib = os.path.join(ROOT, "tuna_amd", "libtunafock.so" if (name == "base" or "=" in name) else f"libtunafock_{name}.so")
        env["TUNAFOCK_LIB"] = lib
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(n)], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:]
        print(name, rep, line, flush=True)
