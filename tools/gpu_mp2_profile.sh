#!/usr/bin/env bash
# GPU box: kernel stats of the AO->MO / RMP2 leg at N = 400 (tools/gpu_mp2_repeat.py under rocprofv3): ms per call of every kernel of the leg
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/mp2prof_${1:-x}"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/tools/gpu_mp2_repeat.py" > "$OUT.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:40]:
    if "eri_team" in r["Name"] or "eri_" in r["Name"][:12]: continue
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"])/1e6:8.3f} ms')
PY
tail -4 "$OUT.log"
