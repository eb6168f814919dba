"""GPU: A/B of the 400-AO SCF leg between library variants (tools/build_variant.sh), each in its own process.
usage: python tools/gpu_scf_ab.py [N] name1 name2 ..."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
args = sys.argv[1:]
n = args.pop(0) if args and args[0].isdigit() else "400"
for rep in range(2):
    for name in args or ["base"]:
        env = dict(os.environ)                               # name = a library variant, or VAR=value for the base library with that variable
        if "=" in name:
            k, v = name.split("=", 1)
            env[k] = v
        env["TUNAFOCK_LIB"] = os.path.join(ROOT, "tuna_amd", "libtunafock.so" if (name == "base" or "=" in name) else f"libtunafock_{name}.so")
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_scf_synth.py"), n, "2"], env=env, capture_output=True, text=True)
        lines = [l for l in out.stdout.splitlines() if l.startswith("synth-")]
        print(name, rep, lines[-1][:300] if lines else out.stderr[-400:], flush=True)
