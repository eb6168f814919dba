"""GPU: A/B timing of the ERI build between library variants (tools/build_variant.sh), each in its own process, alternating;
device time of the ERI kernels from the library's own events (tf_eri_timings), best of five builds.
usage: python tools/gpu_eri_ab.py [N | workload-name] name1 name2 ...   ('base' = tuna_amd/libtunafock.so)"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import bench
from tuna_amd.engine import Engine
atoms, shells, aos, nocc, desc = bench.build_workload(sys.argv[2])
with Engine(0) as eng:
    eng.set_basis(aos)
    best = None
    for rep in range(6):
        t0 = time.perf_counter(); eng.build_eri(True); wall = time.perf_counter() - t0
        t = eng.eri_timings()
        if rep and (best is None or t["cart_kernel_s"] < best["cart_kernel_s"]):
            best = dict(t, wall_s=wall)
    idx = np.random.default_rng(0).integers(0, eng.N, size=(2000, 4)).astype(np.int32)
    print(json.dumps({"eri_kernels_ms": 1e3 * best["cart_kernel_s"], "device_total_ms": 1e3 * best["total_s"], "wall_ms": 1e3 * best["wall_s"],
                      "checksum": float(np.abs(eng.sample_eri(idx)).sum())}))
"""
args = sys.argv[1:]
wl = args.pop(0) if args and (args[0].isdigit() or "-" in args[0]) else "400"
if wl.isdigit():
    wl = "synth-" + wl
names = args or ["base"]
for rep in range(2):
    for name in names:
        # name = library variant, or VAR=value to run the base library with that environment variable (e.g. TF_ERI_FACT_THREADS=256)
        env = dict(os.environ)
        if "=" in name:
            for kv in name.split(","):                                  # VAR=value[,VAR2=value2...]
                k, v = kv.split("=", 1)
                env[k] = v
            lib = os.path.join(ROOT, "tuna_amd", "libtunafock.so")
        else:
            lib = os.path.join(ROOT, "tuna_amd", "libtunafock.so" if name == "base" else f"libtunafock_{name}.so")
        env["TUNAFOCK_LIB"] = lib
        out = subprocess.run([sys.executable, "-c", CHILD, ROOT, wl], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-600:]
        print(name, rep, line, flush=True)
