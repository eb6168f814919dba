#!/usr/bin/env python3
"""Export basis-set tables (pure data) from the reference's tuna_basis.py into JSON.

Runs in the build container only (needs /root/reference).  The reference module is
NOT imported: the dict literals `NAME = { Z : [("S", [(exp, coef), ...]), ...], ... }`
(tuna_basis.py:247-3041) are read with `ast.literal_eval`.  Output:
tuna_amd/data/basis_sets.json  ->  {NAME: {"Z": [[ "S", [[exp, coef], ...]], ...]}}
with the shell order exactly as written in the reference (AO order depends on it,
tuna_molecule.py:553-574).
"""
import ast, json, sys, os

REF = os.environ.get("TUNA_REFERENCE", "/root/reference")
SRC = os.path.join(REF, "TUNA", "tuna_basis.py")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tuna_amd", "data", "basis_sets.json")

WANTED = ["STO_3G", "STO_6G", "_3_21G", "_6_31G", "_6_31GSTAR", "_6_311G", "_6_311GSTARSTAR",
          "CC_PVDZ", "CC_PVTZ", "CC_PVQZ", "CC_PV5Z", "AUG_CC_PVDZ", "AUG_CC_PVTZ", "AUG_CC_PVQZ",
          "DEF2_SVP", "DEF2_TZVP", "DEF2_TZVPP", "DEF2_QZVP"]


def main():
    tree = ast.parse(open(SRC).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
            if name in WANTED:
                table = ast.literal_eval(node.value)
                out[name] = {str(z): [[L, [[float(e), float(c)] for e, c in prims]] for L, prims in shells]
                             for z, shells in table.items()}
    missing = [w for w in WANTED if w not in out]
    if missing:
        sys.exit(f"missing basis tables: {missing}")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", os.path.normpath(OUT), {k: len(v) for k, v in out.items()})


def export_atomic_data():
    """Atomic masses and the spherically averaged HF/STO-3G atomic density matrices of the SAD guess
    (tuna_util.py:1676-1924, pure data) -> tuna_amd/data/atomic_data.json.  The dict literal contains np.array(...) calls,
    so it is evaluated with a namespace that only knows `np.array`."""
    import numpy as np
    lines = open(os.path.join(REF, "TUNA", "tuna_util.py")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("atomic_properties = {"))
    depth, end = 0, start
    for i in range(start, len(lines)):
        depth += lines[i].count("{") - lines[i].count("}")
        if depth == 0:
            end = i
            break
    node = ast.parse("\n".join(lines[start:end + 1])).body[0]
    table = eval(compile(ast.Expression(node.value), "atomic_properties", "eval"), {"np": types_np(np)})
    out = {}
    for sym, d in table.items():
        if sym == "X":
            continue
        out[sym] = {"charge": d["charge"], "mass": d["mass"], "real_vdw_radius": d["real_vdw_radius"], "density": None if d["density"] is None else np.asarray(d["density"], dtype=float).tolist()}
    path = os.path.join(os.path.dirname(OUT), "atomic_data.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", os.path.normpath(path), len(out), "elements")


def types_np(np):
    import types
    return types.SimpleNamespace(array=np.array)


if __name__ == "__main__":
    main()
    export_atomic_data()
