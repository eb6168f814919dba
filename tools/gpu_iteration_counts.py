#!/usr/bin/env python3
"""Where iteration counts may differ from the reference's, and by how much ("Iteration counts at EXTREME thresholds").

(1) the EXTREME keyword runs of tests/golden/keyword_runs.json: our per-iteration table beside the reference's for the last iterations;
(2) the eight finite-field cycles of tests/test_gpu_properties.py run one by one, in host lockstep and in the native batch: iterations per cycle.
Run on the GPU box:  python tools/gpu_iteration_counts.py > gpurun_out/iteration_counts.txt
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLD = os.path.join(ROOT, "tests", "golden")


def keyword_runs():
    from tuna_amd.energy import run
    runs = json.load(open(os.path.join(GOLD, "keyword_runs.json")))
    lines = {"hf_631g": "SPE : F H 0.917 : HF 6-31G : COREGUESS", "co_ccpvdz": "SPE : C O 1.128 : HF CC-PVDZ : COREGUESS"}
    for system in sorted(runs):
        for case in ("diis10", "diis12_damp", "base"):
            g = runs[system]["cases"][case]
            out = run(f"{lines[system]} {g['keywords']}")
            ref = np.array(g["table"])
            ours = np.asarray(out.table)
            print(f"## {system} {case} ({g['keywords']}): iterations ours {out.n_iterations} reference {g['iterations']}; "
                  f"E ours {out.energy:.13f} reference {g['energy']:.13f}")
            n = max(len(ref), len(ours))
            for k in range(max(0, n - 7), n):
                a = ours[k] if k < len(ours) else None
                b = ref[k] if k < len(ref) else None
                fmt = lambda r: "        --        " if r is None else f"dE {r[2]: .3e} rms {r[3]:.2e} max {r[4]:.2e} comm {r[5]:.2e}"
                print(f"  it {k + 1:2d}  ours {fmt(a)}   ref {fmt(b)}")


def field_cycles():
    from test_gpu_properties import FIELDS, _setup
    from tuna_amd import properties as props
    from tuna_amd.engine import Engine
    with Engine(0) as engine:
        g = FIELDS["co_ccpvdz"]
        molecule, calc, integrals, V_NN, X, guess = _setup(engine, g)
        h = g["steps"][1]
        fields = [[0, 0, 2 * h], [0, 0, h], [0, 0, -h], [0, 0, -2 * h], [2 * h, 0, 0], [h, 0, 0], [-h, 0, 0], [-2 * h, 0, 0]]
        print(f"## eight finite-field cycles, CO/cc-pVDZ, thresholds {calc.SCF_conv}")
        res = {}
        for mode in (False, True, "native"):
            per, en = [], []
            for f in fields:
                fe = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched=mode)
                en.append(fe.energies([f])[0])
                per.append(fe.iterations)
            fe = props.FieldEnergies(molecule, calc, integrals, V_NN, X, guess, batched=mode)
            e_all = fe.energies(fields)
            res[mode] = (per, fe.iterations, e_all)
            print(f"  mode {str(mode):7s}: alone, per cycle {per} (sum {sum(per)}); all eight together: {fe.iterations} iterations; "
                  f"max |E_together - E_alone| {np.abs(np.array(e_all) - np.array(en)).max():.2e}")
        base = np.array(res[False][2])
        for mode in (True, "native"):
            print(f"  max |E({mode}) - E(one by one)| {np.abs(np.array(res[mode][2]) - base).max():.2e}")


if __name__ == "__main__":
    keyword_runs()
    field_cycles()
