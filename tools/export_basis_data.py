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


if __name__ == "__main__":
    main()
